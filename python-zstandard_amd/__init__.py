"""MI355X (gfx950) HIP backend for python-zstandard's batch / one-shot frame hot path.

Same Python surface as the reference backend for that path (zstandard/__init__.pyi:87-108, 262-320, 382-442):
``ZstdCompressor.compress`` / ``multi_compress_to_buffer``, ``ZstdDecompressor.decompress`` /
``multi_decompress_to_buffer``, the ``BufferWithSegments*`` types, ``ZstdCompressionDict``, ``ZstdError`` and
``backend_features`` (c-ext/backend_c.c:178-226). Everything else of python-zstandard is out of scope.

The package directory is ``python-zstandard_amd`` (not an identifier); ``import zstandard_amd`` at the repo root is
the importable alias.
"""
from .common import (  # noqa: F401
    BLOCKSIZE_MAX, COMPRESSION_RECOMMENDED_INPUT_SIZE, COMPRESSION_RECOMMENDED_OUTPUT_SIZE, CONTENTSIZE_ERROR, CONTENTSIZE_UNKNOWN,
    DECOMPRESSION_RECOMMENDED_INPUT_SIZE, DECOMPRESSION_RECOMMENDED_OUTPUT_SIZE, DICT_TYPE_AUTO, DICT_TYPE_FULLDICT,
    DICT_TYPE_RAWCONTENT, FORMAT_ZSTD1, FORMAT_ZSTD1_MAGICLESS, MAGIC_NUMBER, MAX_COMPRESSION_LEVEL, WINDOWLOG_MAX, WINDOWLOG_MIN,
    FrameParameters, ZstdCompressionDict, ZstdError, get_frame_parameters,
)
from .buffers import BufferSegment, BufferSegments, BufferWithSegments, BufferWithSegmentsCollection  # noqa: F401
from .compressor import ZstdCompressor  # noqa: F401
from .decompressor import ZstdDecompressor  # noqa: F401
from . import _lib  # noqa: F401

backend = "hip"
backend_features = {"buffer_types", "multi_compress_to_buffer", "multi_decompress_to_buffer"}
ZSTD_VERSION = (1, 5, 7)   # frame bytes match this libzstd release


def frame_content_size(data):
    import numpy as np
    a = np.frombuffer(memoryview(data), dtype=np.uint8)
    v = _lib.lib().zhip_frame_content_size(a.ctypes.data if len(a) else 0, len(a))
    if v == _lib.CONTENTSIZE_ERROR:
        raise ZstdError("error when determining content size")
    return -1 if v == _lib.CONTENTSIZE_UNKNOWN else v


def load_cext():
    """The same surface as a CPython extension (cext/backend_hip.c -> backend_hip.so), like the reference's c-ext backend.
    ``ZSTANDARD_AMD_BACKEND=cext`` makes it the implementation behind this package's names."""
    import importlib
    return importlib.import_module(__name__ + ".backend_hip")


import os as _os
if _os.environ.get("ZSTANDARD_AMD_BACKEND") == "cext":
    _c = load_cext()
    for _n in ("ZstdCompressor", "ZstdDecompressor", "BufferWithSegments", "BufferWithSegmentsCollection", "BufferSegment",
               "BufferSegments", "ZstdCompressionDict", "ZstdError", "frame_content_size", "get_frame_parameters", "FrameParameters"):
        globals()[_n] = getattr(_c, _n)
    backend = _c.backend
