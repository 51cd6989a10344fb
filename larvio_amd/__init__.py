"""larvio_amd — MI355X-native (gfx950) implementation of LARVIO's per-frame hot path.

The product is the C-ABI shared library ``liblvk_hip.so`` (include/lvk_c.h) built from
``larvio_amd/csrc``; this package is the thin Python host side over it (ctypes), mirroring the
reference's ``larvio::ImageProcessor`` / ``larvio::LarVio`` surfaces for tests and benchmarks.
There is no CPU fallback: importing the bindings without the built library, or creating a
context without a GPU, raises.
"""
from ._lib import LvkError, lib, Context  # noqa: F401
from .image_processor import ImageProcessor  # noqa: F401
from .larvio import LarVio  # noqa: F401
