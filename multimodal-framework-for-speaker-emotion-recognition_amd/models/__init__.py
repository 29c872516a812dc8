"""Drop-in mirrors of the reference's ``models`` package for the hot path (reference dir ``model/``, imported as ``models``)."""
