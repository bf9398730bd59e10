"""srmi - MI355X-native super-resolution generator hot path (drop-in for the reference's
`srgan_model.py` / `ops.py` surface).  HIP kernels live in ../csrc, the C ABI in include/srmi.h."""
from . import _lib
from .engine import invalidate  # noqa: F401
from .srgan_model import Discriminator, Generator  # noqa: F401

__all__ = ["Generator", "Discriminator", "invalidate"]
