"""MI355X (gfx950) HIP backend for python-zstandard's batch / one-shot frame hot path.

Same Python surface as the reference backend for that path (zstandard/__init__.pyi:87-108, 262-320, 382-442):
``ZstdCompressor.compress`` / ``multi_compress_to_buffer``, ``ZstdDecompressor.decompress`` / ``multi_decompress_to_buffer``,
``ZstdCompressionParameters``, the ``BufferWithSegments*`` types, ``ZstdCompressionDict``, ``ZstdError`` and ``backend_features``
(c-ext/backend_c.c:178-226). Everything else of python-zstandard is out of scope.

The host side is what it is in the reference: a CPython extension written in C (``cext/backend_hip.c`` -> ``backend_hip.so``)
over the C ABI of ``csrc/libzstd_hip.so`` (``include/zstd_hip.h``). This module only re-exports it and adds the pieces that have no
counterpart in the reference: ``device`` (HBM-resident batches as torch tensors), ``sharded`` / ``parallel`` (one process per GPU).
There is no CPU fallback and no second implementation: a missing extension or library fails the import.

The package directory is ``python-zstandard_amd`` (not an identifier); ``import zstandard_amd`` at the repo root is the alias.
"""
import os as _os

try:
    from . import backend_hip as _c
except ImportError as _e:                                   # loud, with the fix
    raise ImportError("the HIP backend's C extension is not built (%s): run `python __graft_entry__.py` "
                      "(python-zstandard_amd/csrc/build.sh + cext/build.sh); there is no CPU fallback" % (_e,))

for _n in dir(_c):
    if not _n.startswith("_"):
        globals()[_n] = getattr(_c, _n)
del _n

from . import _lib  # noqa: E402,F401   ctypes view of the same library (device-resident API, tests)

backend = "hip"                    # the reference's `zstandard.backend`; backend_hip.backend says "hip_cext"


def compress(data, level=3):
    """one-shot convenience of the reference package (zstandard/__init__.py:184-199): ``ZstdCompressor(level=level).compress(data)``"""
    return ZstdCompressor(level=level).compress(data)          # noqa: F821  (re-exported from the extension above)


def decompress(data, max_output_size=0):
    """zstandard/__init__.py:202-217: ``ZstdDecompressor().decompress(data, max_output_size=max_output_size)``"""
    return ZstdDecompressor().decompress(data, max_output_size=max_output_size)          # noqa: F821


def load_cext():
    """the extension module itself (kept for callers written against round 1, where it was optional)"""
    return _c
