"""mser: Python binding of libmser.so (HIP kernels for gfx950) + host-side composition of the speaker-aware LSTHM path."""
from . import _lib

__all__ = ["_lib"]
