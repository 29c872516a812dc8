"""Drop-in mirror of the reference's ``attention`` package for the one module the hot path names (SelfAttention)."""
