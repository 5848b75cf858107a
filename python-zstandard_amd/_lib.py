"""ctypes binding of csrc/libzstd_hip.so (C ABI declared in include/zstd_hip.h).

There is NO fallback: if the HIP library is missing or no GPU is visible, the calls that need it raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZHIP_LIB") or os.path.join(_HERE, "csrc", "libzstd_hip.so")

ERR_NONE, ERR_ZSTD, ERR_NO_MEMORY, ERR_SIZE_MISMATCH, ERR_UNKNOWN_SIZE, ERR_HIP, ERR_UNSUPPORTED = range(7)
CONTENTSIZE_UNKNOWN = 2**64 - 1
CONTENTSIZE_ERROR = 2**64 - 2
FLAG_ALLOW_SHORT = 2   # zhip_decompress_batch requireSizes bit: dstSize is a capacity, not an exact size


class Segment(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint64)]


class Item(C.Structure):
    _fields_ = [("src", C.c_void_p), ("srcSize", C.c_size_t), ("dstSize", C.c_size_t)]


class OutBuf(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dataSize", C.c_size_t), ("segs", C.POINTER(Segment)), ("nSegs", C.c_size_t)]


class Error(C.Structure):
    _fields_ = [("kind", C.c_int), ("zstdErr", C.c_int), ("index", C.c_size_t), ("detail", C.c_uint64 * 2)]


class CompressionParameters(C.Structure):
    _fields_ = [("windowLog", C.c_uint32), ("chainLog", C.c_uint32), ("hashLog", C.c_uint32), ("searchLog", C.c_uint32),
                ("minMatch", C.c_uint32), ("targetLength", C.c_uint32), ("strategy", C.c_int32)]


class CParams(C.Structure):
    _fields_ = [("level", C.c_int), ("contentSizeFlag", C.c_int), ("checksumFlag", C.c_int), ("dictIDFlag", C.c_int),
                ("dict", C.c_void_p), ("dictSize", C.c_size_t), ("dictType", C.c_int), ("format", C.c_int),
                ("cp", CompressionParameters)]


class DParams(C.Structure):
    _fields_ = [("dict", C.c_void_p), ("dictSize", C.c_size_t), ("maxWindowSize", C.c_uint64), ("dictType", C.c_int),
                ("format", C.c_int)]


DICT_AUTO, DICT_RAWCONTENT, DICT_FULLDICT = 0, 1, 2
FORMAT_ZSTD1, FORMAT_ZSTD1_MAGICLESS = 0, 1


_lib = None


def lib():
    """Load libzstd_hip.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libzstd_hip.so is not built (%s). Run `python __graft_entry__.py` / "
            "`python-zstandard_amd/csrc/build.sh`; this backend has no CPU fallback." % LIB_PATH)
    try:
        # torch bundles its own libamdhip64.so.7; loading it first makes this library bind to the SAME HIP runtime
        # instance, which is required for torch streams / device pointers to be meaningful to our launches.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    protos = {
        "zhip_abi_version": (C.c_int, []),
        "zhip_device_count": (C.c_int, []),
        "zhip_set_device": (C.c_int, [C.c_int]),
        "zhip_last_error": (C.c_char_p, []),
        "zhip_error_name": (C.c_char_p, [C.c_int]),
        "zhip_selftest": (C.c_int, []),
        "zhip_compress_bound": (sz, [sz]),
        "zhip_frame_content_size": (u64, [vp, sz]),
        "zhip_find_frame_compressed_size": (C.c_int64, [vp, sz]),
        "zhip_frame_content_size_format": (u64, [vp, sz, C.c_int]),
        "zhip_find_frame_compressed_size_format": (C.c_int64, [vp, sz, C.c_int]),
        "zhip_get_cparams": (None, [C.c_int, u64, sz, C.POINTER(CompressionParameters)]),
        "zhip_thread_memory_size": (sz, []),
        "zhip_compress_batch": (C.c_int, [C.POINTER(CParams), C.POINTER(Item), sz, C.POINTER(C.POINTER(OutBuf)),
                                          C.POINTER(sz), C.POINTER(Error)]),
        "zhip_decompress_batch": (C.c_int, [C.POINTER(DParams), C.POINTER(Item), sz, C.c_int,
                                            C.POINTER(C.POINTER(OutBuf)), C.POINTER(sz), C.POINTER(Error)]),
        "zhip_free_outbufs": (None, [C.POINTER(OutBuf), sz, C.c_int]),
        "zhip_free_payload": (None, [C.c_void_p]),
        "zhip_ctx_create": (vp, []),
        "zhip_ctx_destroy": (None, [vp]),
        "zhip_ctx_set_ddict": (C.c_int, [vp, vp, sz, C.c_int]),
        "zhip_ctx_set_dformat": (C.c_int, [vp, C.c_int, u64]),
        "zhip_ctx_set_cparams": (C.c_int, [vp, C.POINTER(CParams)]),
        "zhip_decompress_batch_device": (C.c_int, [vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "zhip_compress_batch_device": (C.c_int, [vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "zhip_ctx_sync": (C.c_int, [vp, vp, vp, sz, C.POINTER(Error)]),
        "zhip_compact_device": (C.c_int, [vp, vp, vp, vp, vp, sz, vp, vp]),
        "zhip_ctx_set_size_hint": (None, [vp, u64]),
        "zhip_kernel_name": (C.c_char_p, [C.c_int]),
        "zhip_ctx_kernel_time": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]),
        "zhip_ctx_table_pick": (C.c_int, [vp, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in protos.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if L.zhip_abi_version() != 3:
        raise ImportError("libzstd_hip.so ABI mismatch")
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "zhip_abi_version", "zhip_device_count", "zhip_set_device", "zhip_last_error", "zhip_error_name",
    "zhip_selftest", "zhip_compress_bound", "zhip_frame_content_size", "zhip_find_frame_compressed_size", "zhip_frame_content_size_format",
    "zhip_find_frame_compressed_size_format", "zhip_get_cparams", "zhip_ctx_set_dformat", "zhip_compress_batch",
    "zhip_decompress_batch", "zhip_free_outbufs", "zhip_free_payload", "zhip_ctx_create", "zhip_ctx_destroy", "zhip_ctx_set_ddict",
    "zhip_ctx_set_cparams", "zhip_decompress_batch_device", "zhip_compress_batch_device", "zhip_ctx_sync",
    "zhip_kernel_name", "zhip_ctx_kernel_time", "zhip_thread_memory_size", "zhip_compact_device", "zhip_ctx_set_size_hint",
    "zhip_ctx_table_pick", "zhip_partition_by_bytes", "zhip_batch_devices",
]


def error_name(code):
    return lib().zhip_error_name(code).decode()


def last_error():
    return lib().zhip_last_error().decode()
