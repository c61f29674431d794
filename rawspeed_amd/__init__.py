"""rawspeed_amd -- MI355X-native RAW decompression core behind rawspeed's
decompressor API (UncompressedDecompressor / LJpegDecompressor /
Cr2Decompressor / AbstractDngDecompressor).

The product is the C-ABI shared library rawspeed_amd/librsx.so (include/rsx.h):
hand-written HIP kernels for gfx950 plus the host glue.  This Python package is
a thin ctypes front-end used by tests and bench.py; it holds no algorithm.
"""
from . import abi  # noqa: F401

__version__ = "0.1.0"
