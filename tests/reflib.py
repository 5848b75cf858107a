"""ctypes bindings for the two CHECKERS (test infrastructure, never imported by the product package):

* ``RefZstd``  -- oracle/_ref/libzstd_ref.so, the reference's vendored libzstd 1.5.7 compiled from
  /root/reference/zstd/zstd.c by oracle/Makefile.  Driven exactly like the reference's C extension drives it
  (c-ext/compressor.c:209-233 parameters, :1035-1043 ZSTD_CCtx_setPledgedSrcSize + ZSTD_compressStream2(e_end);
  c-ext/decompressor.c:1150 ZSTD_decompressStream), so its frames ARE the reference's frames.
* ``Oracle``   -- oracle/libzstd_oracle.so, our plain-C restatement (oracle/zo_*.c).
"""
import ctypes as C
import hashlib
import os
import threading

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libzstd_ref.so")
ORACLE_SO = os.path.join(_ROOT, "oracle", "libzstd_oracle.so")

F_CONTENTSIZE, F_CHECKSUM, F_DICTID = 1, 2, 4
DEFAULT_FLAGS = F_CONTENTSIZE | F_DICTID  # python-zstandard defaults (c-ext/compressor.c:209-233)

# ZSTD_cParameter values (zstd.h)
_P_LEVEL, _P_CONTENTSIZE, _P_CHECKSUM, _P_DICTID = 100, 200, 201, 202


class _Buf(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class _CParams(C.Structure):      # ZSTD_compressionParameters (zstd.h)
    _fields_ = [("windowLog", C.c_uint), ("chainLog", C.c_uint), ("hashLog", C.c_uint), ("searchLog", C.c_uint), ("minMatch", C.c_uint),
                ("targetLength", C.c_uint), ("strategy", C.c_int)]


class _CMem(C.Structure):         # ZSTD_customMem: all NULL = the default allocator
    _fields_ = [("customAlloc", C.c_void_p), ("customFree", C.c_void_p), ("opaque", C.c_void_p)]


REF_KIND = None     # which libzstd 1.5.7 the checker is: "reference build" (oracle/_ref, compiled from /root/reference/zstd/zstd.c) or "image copy"


def _image_libzstd():
    """the stock libzstd 1.5.7 inside this image (byte-identical level-3 frames to the reference build)"""
    import glob
    cands = sorted(glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so.1.5.7"))
    return cands[0] if cands else None


def have_ref():
    """True when a REAL libzstd 1.5.7 can be the checker: the reference build (oracle/_ref/libzstd_ref.so, git-ignored: it reaches the GPU
    box only with the working tree) or, failing that, the image's own copy. Never the restatement (a clone-based run must not
    quietly compare the kernels with builder-written code)."""
    global REF_SO, REF_KIND
    if os.path.exists(os.path.join(_ROOT, "oracle", "_ref", "libzstd_ref.so")):
        REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libzstd_ref.so"); REF_KIND = "reference build (oracle/_ref)"
        return True
    alt = _image_libzstd()
    if alt:
        REF_SO = alt; REF_KIND = "image copy (%s)" % alt
        return True
    return False


def checker():
    """the GPU suites' encoder / checker: libzstd 1.5.7 itself, or an error -- not a skip, not the restatement"""
    if not have_ref():
        raise RuntimeError("no libzstd 1.5.7 to check against: neither oracle/_ref/libzstd_ref.so (make -C oracle ref) nor the image's copy")
    return RefZstd()


def have_oracle():
    return os.path.exists(ORACLE_SO)


class RefZstd:
    def __init__(self):
        self.lib = L = C.CDLL(REF_SO)
        for name, res, args in [
            ("ZSTD_versionNumber", C.c_uint, []),
            ("ZSTD_createCCtx", C.c_void_p, []),
            ("ZSTD_freeCCtx", C.c_size_t, [C.c_void_p]),
            ("ZSTD_CCtx_setParameter", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
            ("ZSTD_CCtx_setPledgedSrcSize", C.c_size_t, [C.c_void_p, C.c_ulonglong]),
            ("ZSTD_CCtx_loadDictionary", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
            ("ZSTD_CCtx_loadDictionary_advanced", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int]),
            ("ZSTD_DCtx_loadDictionary_advanced", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int]),
            ("ZSTD_DCtx_setParameter", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
            ("ZSTD_CCtx_reset", C.c_size_t, [C.c_void_p, C.c_int]),
            ("ZSTD_compressStream2", C.c_size_t, [C.c_void_p, C.POINTER(_Buf), C.POINTER(_Buf), C.c_int]),
            ("ZSTD_compressBound", C.c_size_t, [C.c_size_t]),
            ("ZSTD_createDCtx", C.c_void_p, []),
            ("ZSTD_freeDCtx", C.c_size_t, [C.c_void_p]),
            ("ZSTD_DCtx_loadDictionary", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
            ("ZSTD_decompressStream", C.c_size_t, [C.c_void_p, C.POINTER(_Buf), C.POINTER(_Buf)]),
            ("ZSTD_DCtx_reset", C.c_size_t, [C.c_void_p, C.c_int]),
            ("ZSTD_isError", C.c_uint, [C.c_size_t]),
            ("ZSTD_getErrorName", C.c_char_p, [C.c_size_t]),
            ("ZSTD_getFrameContentSize", C.c_ulonglong, [C.c_void_p, C.c_size_t]),
            ("ZDICT_trainFromBuffer", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_uint]),
            ("ZSTD_getCParams", _CParams, [C.c_int, C.c_ulonglong, C.c_size_t]),
            ("ZSTD_createCDict_advanced", C.c_void_p, [C.c_char_p, C.c_size_t, C.c_int, C.c_int, _CParams, _CMem]),
            ("ZSTD_freeCDict", C.c_size_t, [C.c_void_p]),
            ("ZSTD_CCtx_refCDict", C.c_size_t, [C.c_void_p, C.c_void_p]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        assert L.ZSTD_versionNumber() == 10507, "oracle/_ref must be libzstd 1.5.7"
        self._tls = threading.local()

    # -- contexts are per-thread, reused across frames like compress_worker does (compressor.c:1129-1168)
    # ZSTD_cParameter numbers (zstd.h): what c-ext/compressionparams.c sets from a ZstdCompressionParameters object
    PARAM_IDS = {"window_log": 101, "hash_log": 102, "chain_log": 103, "search_log": 104, "min_match": 105, "target_length": 106,
                 "strategy": 107, "format": 10}

    def compress_advanced(self, data, level=3, flags=DEFAULT_FLAGS, dict_data=None, dict_type=0, **params):
        """one frame with explicit parameters / dictionary content type / frame format, driven like compress_worker drives a context
        whose ZSTD_CCtx_params came from a ZstdCompressionParameters object (a fresh context per call)"""
        L = self.lib
        data = bytes(data)
        ctx = L.ZSTD_createCCtx()
        try:
            L.ZSTD_CCtx_setParameter(ctx, _P_LEVEL, level)
            L.ZSTD_CCtx_setParameter(ctx, _P_CONTENTSIZE, 1 if flags & F_CONTENTSIZE else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_CHECKSUM, 1 if flags & F_CHECKSUM else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_DICTID, 1 if flags & F_DICTID else 0)
            for k, v in params.items():
                r = L.ZSTD_CCtx_setParameter(ctx, self.PARAM_IDS[k], v)
                if L.ZSTD_isError(r):
                    raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            if dict_data:
                r = L.ZSTD_CCtx_loadDictionary_advanced(ctx, dict_data, len(dict_data), 1, dict_type)     # ZSTD_dlm_byRef
                if L.ZSTD_isError(r):
                    raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            L.ZSTD_CCtx_setPledgedSrcSize(ctx, len(data))
            cap = L.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(max(cap, 1))
            src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
            out = _Buf(C.addressof(dst), cap, 0)
            inp = _Buf(C.addressof(src), len(data), 0)
            r = L.ZSTD_compressStream2(ctx, C.byref(out), C.byref(inp), 2)
            if L.ZSTD_isError(r):
                raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            assert r == 0
            return dst.raw[: out.pos]
        finally:
            L.ZSTD_freeCCtx(ctx)

    def decompress_advanced(self, frame, out_size, dict_data=None, dict_type=0, format=0):
        L = self.lib
        ctx = L.ZSTD_createDCtx()
        try:
            L.ZSTD_DCtx_setParameter(ctx, 1000, format)                         # ZSTD_d_format (experimentalParam1)
            if dict_data:
                r = L.ZSTD_DCtx_loadDictionary_advanced(ctx, dict_data, len(dict_data), 1, dict_type)
                if L.ZSTD_isError(r):
                    raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            frame = bytes(frame)
            dst = C.create_string_buffer(max(out_size, 1))
            src = C.create_string_buffer(frame, len(frame))
            out = _Buf(C.addressof(dst), out_size, 0)
            inp = _Buf(C.addressof(src), len(frame), 0)
            r = L.ZSTD_decompressStream(ctx, C.byref(out), C.byref(inp))
            if L.ZSTD_isError(r):
                raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            return dst.raw[: out.pos]
        finally:
            L.ZSTD_freeDCtx(ctx)

    def cdict_params(self, level, dict_size):
        """ZSTD_getCParams(level, 0, dictSize) as a dict of ZstdCompressionParameters-style keys (what precompute_compress(level=) starts from)"""
        cp = self.lib.ZSTD_getCParams(level, 0, dict_size)
        return dict(window_log=cp.windowLog, chain_log=cp.chainLog, hash_log=cp.hashLog, search_log=cp.searchLog, min_match=cp.minMatch,
                    target_length=cp.targetLength, strategy=cp.strategy)

    def compress_with_cdict(self, data, dict_data, level=3, flags=DEFAULT_FLAGS, cdict_level=0, cdict_params=None, dict_type=0, window_log=0):
        """one frame through a PRECOMPUTED dictionary, the way the reference does it after ZstdCompressionDict.precompute_compress
        (c-ext/compressiondict.c:266-278 ZSTD_getCParams / to_cparams + ZSTD_createCDict_advanced, c-ext/compressor.c:29-31 ZSTD_CCtx_refCDict);
        cdict_params: the seven ZSTD_compressionParameters fields by their ZstdCompressionParameters names (0 = unset)"""
        L = self.lib
        data = bytes(data)
        if cdict_params is None:
            cp = L.ZSTD_getCParams(cdict_level, 0, len(dict_data))
        else:
            q = cdict_params
            cp = _CParams(q.get("window_log", 0), q.get("chain_log", 0), q.get("hash_log", 0), q.get("search_log", 0), q.get("min_match", 0),
                          q.get("target_length", 0), q.get("strategy", 0))
        cd = L.ZSTD_createCDict_advanced(dict_data, len(dict_data), 1, dict_type, cp, _CMem(None, None, None))
        if not cd:
            raise RuntimeError("unable to precompute dictionary")
        ctx = L.ZSTD_createCCtx()
        try:
            L.ZSTD_CCtx_setParameter(ctx, _P_LEVEL, level)
            L.ZSTD_CCtx_setParameter(ctx, _P_CONTENTSIZE, 1 if flags & F_CONTENTSIZE else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_CHECKSUM, 1 if flags & F_CHECKSUM else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_DICTID, 1 if flags & F_DICTID else 0)
            if window_log:
                L.ZSTD_CCtx_setParameter(ctx, self.PARAM_IDS["window_log"], window_log)
            r = L.ZSTD_CCtx_refCDict(ctx, cd)
            if L.ZSTD_isError(r):
                raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            L.ZSTD_CCtx_setPledgedSrcSize(ctx, len(data))
            cap = L.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(max(cap, 1))
            src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
            out = _Buf(C.addressof(dst), cap, 0)
            inp = _Buf(C.addressof(src), len(data), 0)
            r = L.ZSTD_compressStream2(ctx, C.byref(out), C.byref(inp), 2)
            if L.ZSTD_isError(r):
                raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            assert r == 0
            return dst.raw[: out.pos]
        finally:
            L.ZSTD_freeCCtx(ctx)
            L.ZSTD_freeCDict(cd)

    def _cctx(self, level, flags, dict_data):
        # keyed by the dictionary's CONTENT: id() of a freed bytes object is handed out again, and a cached context would then carry
        # the wrong dictionary (seen as a rare order-dependent mismatch in the suite)
        key = (level, flags, hashlib.sha256(dict_data).digest() if dict_data else None)
        cache = getattr(self._tls, "cctx", None)
        if cache is None:
            cache = self._tls.cctx = {}
        ctx = cache.get(key)
        if ctx is None:
            L = self.lib
            ctx = L.ZSTD_createCCtx()
            L.ZSTD_CCtx_setParameter(ctx, _P_LEVEL, level)
            L.ZSTD_CCtx_setParameter(ctx, _P_CONTENTSIZE, 1 if flags & F_CONTENTSIZE else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_CHECKSUM, 1 if flags & F_CHECKSUM else 0)
            L.ZSTD_CCtx_setParameter(ctx, _P_DICTID, 1 if flags & F_DICTID else 0)
            if dict_data:
                r = L.ZSTD_CCtx_loadDictionary(ctx, dict_data, len(dict_data))
                assert not L.ZSTD_isError(r)
            cache[key] = ctx
        return ctx

    def compress_into(self, dst_addr, dst_cap, src_addr, src_size, level=3, flags=DEFAULT_FLAGS, dict_data=None):
        L = self.lib
        ctx = self._cctx(level, flags, dict_data)
        L.ZSTD_CCtx_setPledgedSrcSize(ctx, src_size)
        out = _Buf(dst_addr, dst_cap, 0)
        inp = _Buf(src_addr, src_size, 0)
        r = L.ZSTD_compressStream2(ctx, C.byref(out), C.byref(inp), 2)  # ZSTD_e_end
        if L.ZSTD_isError(r):
            L.ZSTD_CCtx_reset(ctx, 1)
            raise RuntimeError(L.ZSTD_getErrorName(r).decode())
        assert r == 0
        return out.pos

    def compress(self, data, level=3, flags=DEFAULT_FLAGS, dict_data=None):
        data = bytes(data)
        cap = self.lib.ZSTD_compressBound(len(data))
        dst = C.create_string_buffer(max(cap, 1))
        src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
        n = self.compress_into(C.addressof(dst), cap, C.addressof(src), len(data), level, flags, dict_data)
        return dst.raw[:n]

    def decompress(self, frame, out_size, dict_data=None):
        L = self.lib
        ctx = L.ZSTD_createDCtx()
        try:
            if dict_data:
                L.ZSTD_DCtx_loadDictionary(ctx, dict_data, len(dict_data))
            frame = bytes(frame)
            dst = C.create_string_buffer(max(out_size, 1))
            src = C.create_string_buffer(frame, len(frame))
            out = _Buf(C.addressof(dst), out_size, 0)
            inp = _Buf(C.addressof(src), len(frame), 0)
            r = L.ZSTD_decompressStream(ctx, C.byref(out), C.byref(inp))
            if L.ZSTD_isError(r):
                raise RuntimeError(L.ZSTD_getErrorName(r).decode())
            if r != 0:
                raise RuntimeError("partial frame (hint %d)" % r)
            return dst.raw[: out.pos]
        finally:
            L.ZSTD_freeDCtx(ctx)

    def decompress_into(self, dctx, dst_addr, dst_cap, src_addr, src_size):
        out = _Buf(dst_addr, dst_cap, 0)
        inp = _Buf(src_addr, src_size, 0)
        r = self.lib.ZSTD_decompressStream(dctx, C.byref(out), C.byref(inp))
        if r != 0:
            raise RuntimeError("ref decompress failed: %r" % r)
        return out.pos

    def frame_content_size(self, frame):
        frame = bytes(frame)
        return self.lib.ZSTD_getFrameContentSize(frame, len(frame))

    def train_dictionary(self, dict_size, samples):
        blob = b"".join(samples)
        sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
        dst = C.create_string_buffer(dict_size)
        r = self.lib.ZDICT_trainFromBuffer(dst, dict_size, blob, sizes, len(samples))
        if self.lib.ZSTD_isError(r):
            raise RuntimeError(self.lib.ZSTD_getErrorName(r).decode())
        return dst.raw[:r]


class Oracle:
    def __init__(self):
        self.lib = L = C.CDLL(ORACLE_SO)
        L.zo_compress_bound.restype = C.c_size_t
        L.zo_compress_bound.argtypes = [C.c_size_t]
        L.zo_compress_frame.restype = C.c_int64
        L.zo_compress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint,
                                        C.c_char_p, C.c_size_t]
        L.zo_decompress_frame.restype = C.c_int64
        L.zo_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.zo_frame_content_size.restype = C.c_uint64
        L.zo_frame_content_size.argtypes = [C.c_char_p, C.c_size_t]
        L.zo_find_frame_compressed_size.restype = C.c_int64
        L.zo_find_frame_compressed_size.argtypes = [C.c_char_p, C.c_size_t]
        L.zo_xxh64.restype = C.c_uint64
        L.zo_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]

    def compress_into(self, dst_addr, dst_cap, src_addr, src_size, level=3, flags=DEFAULT_FLAGS, dict_data=None):
        r = self.lib.zo_compress_frame(dst_addr, dst_cap, src_addr, src_size, level, flags, dict_data,
                                       len(dict_data) if dict_data else 0)
        if r < 0:
            raise RuntimeError("oracle compress error %d" % -r)
        return r

    def compress(self, data, level=3, flags=DEFAULT_FLAGS, dict_data=None):
        data = bytes(data)
        cap = self.lib.zo_compress_bound(len(data))
        dst = C.create_string_buffer(max(cap, 1))
        src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
        n = self.compress_into(C.addressof(dst), cap, C.addressof(src), len(data), level, flags, dict_data)
        return dst.raw[:n]

    def decompress_into(self, dst_addr, dst_cap, src_addr, src_size, dict_data=None):
        return self.lib.zo_decompress_frame(dst_addr, dst_cap, src_addr, src_size, dict_data,
                                            len(dict_data) if dict_data else 0, None)

    def decompress(self, frame, out_size, dict_data=None):
        frame = bytes(frame)
        dst = C.create_string_buffer(max(out_size, 1))
        src = C.create_string_buffer(frame, len(frame))
        r = self.decompress_into(C.addressof(dst), out_size, C.addressof(src), len(frame), dict_data)
        if r < 0:
            raise RuntimeError("oracle decompress error %d" % -r)
        return dst.raw[:r]

    def frame_content_size(self, frame):
        frame = bytes(frame)
        return self.lib.zo_frame_content_size(frame, len(frame))

    def xxh64(self, data, seed=0):
        data = bytes(data)
        return self.lib.zo_xxh64(data, len(data), seed)
