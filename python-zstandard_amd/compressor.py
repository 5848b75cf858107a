"""ZstdCompressor: host-side mirror of c-ext/compressor.c for the hot path only
(``__init__`` :88-246, ``compress`` :509-574, ``multi_compress_to_buffer`` :1340-1503); frames are built by HIP kernels
and are bit-identical to libzstd 1.5.7 for the supported strategies.
"""
import ctypes as C

import numpy as np

from . import _lib
from .buffers import BufferWithSegments, BufferWithSegmentsCollection
from .common import MAX_COMPRESSION_LEVEL, ZstdCompressionDict, ZstdError, collect_sources


class ZstdCompressor:
    def __init__(self, level=3, dict_data=None, compression_params=None, write_checksum=None,
                 write_content_size=None, write_dict_id=None, threads=0):
        if level > MAX_COMPRESSION_LEVEL:
            raise ValueError("level must be less than %d" % (MAX_COMPRESSION_LEVEL + 1))
        if threads < 0:
            threads = 0
        if compression_params is not None:
            # the reference's mutual-exclusion checks come first (compressor.c:177-200) ...
            for given, name in ((write_checksum, "write_checksum"), (write_content_size, "write_content_size"),
                                (write_dict_id, "write_dict_id"), (threads or None, "threads")):
                if given is not None:
                    raise ValueError("cannot define compression_params and %s" % name)
            # ... explicit compression parameters themselves are not plumbed to the kernels yet: fail loudly
            raise ZstdError("compression_params is not supported by the HIP backend; pass level=")
        if dict_data is not None and not isinstance(dict_data, ZstdCompressionDict):
            raise TypeError("dict_data must be a ZstdCompressionDict")
        self._level = level
        self._dict = dict_data
        self._write_checksum = bool(write_checksum) if write_checksum is not None else False
        self._write_content_size = bool(write_content_size) if write_content_size is not None else True
        self._write_dict_id = bool(write_dict_id) if write_dict_id is not None else True

    def memory_size(self):
        return 0

    def _cparams(self):
        p = _lib.CParams()
        p.level = self._level
        p.contentSizeFlag = int(self._write_content_size)
        p.checksumFlag = int(self._write_checksum)
        p.dictIDFlag = int(self._write_dict_id)
        self._dict_keep = None
        if self._dict is not None and len(self._dict):
            raw = self._dict.as_bytes()
            self._dict_keep = np.frombuffer(raw, dtype=np.uint8)
            p.dict = self._dict_keep.ctypes.data
            p.dictSize = len(raw)
        return p

    def _run(self, views):
        L = _lib.lib()
        n = len(views)
        items = (_lib.Item * n)()
        keep = []
        for i, mv in enumerate(views):
            a = np.frombuffer(mv, dtype=np.uint8)
            keep.append(a)
            items[i].src = a.ctypes.data if len(a) else 0
            items[i].srcSize = len(a)
        out = C.POINTER(_lib.OutBuf)()
        n_out = C.c_size_t(0)
        err = _lib.Error()
        params = self._cparams()
        rc = L.zhip_compress_batch(C.byref(params), items, n, C.byref(out), C.byref(n_out), C.byref(err))
        return rc, err, out, n_out.value

    def _raise(self, rc, err, one_shot):
        if rc == _lib.ERR_ZSTD:
            name = _lib.error_name(err.zstdErr)
            raise ZstdError(("cannot compress: %s" % name) if one_shot else ("error compressing item %d: %s" % (err.index, name)))
        if rc == _lib.ERR_NO_MEMORY:
            raise MemoryError()
        if rc == _lib.ERR_SIZE_MISMATCH:
            raise ZstdError("error compressing item %d: not enough space in output" % err.index)
        raise ZstdError("HIP backend failure: %s" % _lib.last_error())

    def compress(self, data):
        mv = memoryview(data)
        if not mv.c_contiguous:
            raise ValueError("data buffer should be contiguous and have at most one dimension")
        mv = mv.cast("B") if (mv.format != "B" or mv.ndim != 1) else mv
        rc, err, out, n_out = self._run([mv])
        if rc != _lib.ERR_NONE:
            self._raise(rc, err, True)
        L = _lib.lib()
        try:
            ob = out[0]
            return C.string_at(ob.data + ob.segs[0].offset, ob.segs[0].length)
        finally:
            L.zhip_free_outbufs(out, n_out, 1)

    def multi_compress_to_buffer(self, data, threads=0):
        """``threads`` is accepted for API compatibility (compressor.c:1361-1367); the GPU does the fan-out."""
        views = collect_sources(data, "argument must be list of BufferWithSegments")
        if not views:
            raise ValueError("no source elements found")
        if sum(len(v) for v in views) == 0:
            raise ValueError("source elements are empty")
        rc, err, out, n_out = self._run(views)
        if rc != _lib.ERR_NONE:
            self._raise(rc, err, False)
        L = _lib.lib()
        buffers = []
        for i in range(n_out):
            ob = out[i]
            buffers.append(BufferWithSegments._from_memory(ob.data, ob.dataSize, C.cast(ob.segs, C.c_void_p).value, ob.nSegs))
        L.zhip_free_outbufs(out, n_out, 0)
        return BufferWithSegmentsCollection(*buffers)
