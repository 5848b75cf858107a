"""ZstdDecompressor: host-side mirror of c-ext/decompressor.c for the hot path only
(``decompress`` :263-395, ``multi_decompress_to_buffer`` :1459-1710); the frame loop runs in HIP kernels.
"""
import ctypes as C
import struct

import numpy as np

from . import _lib
from .buffers import BufferWithSegments, BufferWithSegmentsCollection, _addr_of
from .common import FORMAT_ZSTD1, ZstdCompressionDict, ZstdError, collect_sources


class ZstdDecompressor:
    def __init__(self, dict_data=None, max_window_size=0, format=FORMAT_ZSTD1):
        if dict_data is not None and not isinstance(dict_data, ZstdCompressionDict):
            raise TypeError("dict_data must be a ZstdCompressionDict")
        if format != FORMAT_ZSTD1:
            raise ZstdError("unable to set decoding format: only FORMAT_ZSTD1 is supported by the HIP backend")
        self._dict = dict_data
        self._max_window_size = max_window_size
        self._format = format

    def memory_size(self):
        return 0

    # ------------------------------------------------------------------------------------------------ helpers
    def _dparams(self):
        p = _lib.DParams()
        self._dict_keep = None
        if self._dict is not None and len(self._dict):
            raw = self._dict.as_bytes()
            self._dict_keep = np.frombuffer(raw, dtype=np.uint8)
            p.dict = self._dict_keep.ctypes.data
            p.dictSize = len(raw)
        p.maxWindowSize = self._max_window_size or 0
        return p

    def _run(self, views, dst_sizes, flags):
        L = _lib.lib()
        n = len(views)
        items = (_lib.Item * n)()
        keep = []
        for i, mv in enumerate(views):
            a = np.frombuffer(mv, dtype=np.uint8)
            keep.append(a)
            items[i].src = a.ctypes.data if len(a) else 0
            items[i].srcSize = len(a)
            items[i].dstSize = dst_sizes[i] if dst_sizes is not None else 0
        out = C.POINTER(_lib.OutBuf)()
        n_out = C.c_size_t(0)
        err = _lib.Error()
        params = self._dparams()
        rc = L.zhip_decompress_batch(C.byref(params), items, n, flags, C.byref(out), C.byref(n_out), C.byref(err))
        return rc, err, out, n_out.value

    # ------------------------------------------------------------------------------------------------ one-shot
    def decompress(self, data, max_output_size=0, read_across_frames=False, allow_extra_data=True):
        if read_across_frames:
            raise ZstdError("ZstdDecompressor.read_across_frames=True is not yet implemented")
        L = _lib.lib()
        mv = memoryview(data)
        if not mv.c_contiguous:
            raise ValueError("data buffer should be contiguous and have at most one dimension")
        mv = mv.cast("B") if (mv.format != "B" or mv.ndim != 1) else mv
        src = np.frombuffer(mv, dtype=np.uint8)
        addr = src.ctypes.data if len(src) else 0
        fcs = L.zhip_frame_content_size(addr, len(src))
        if fcs == _lib.CONTENTSIZE_ERROR:
            raise ZstdError("error determining content size from frame header")
        if fcs == 0:
            return b""
        flags = 0
        if fcs == _lib.CONTENTSIZE_UNKNOWN:
            if max_output_size == 0:
                raise ZstdError("could not determine content size in frame header")
            if max_output_size > 2**48:
                raise MemoryError()
            cap, expected, flags = max_output_size, 0, _lib.FLAG_ALLOW_SHORT
        else:
            cap, expected = fcs, fcs
        rc, err, out, n_out = self._run([mv], [cap], flags)
        if rc == _lib.ERR_ZSTD:
            if flags and err.zstdErr == 70:   # destination full before the frame ended (streaming hint != 0)
                raise ZstdError("decompression error: did not decompress full frame")
            raise ZstdError("decompression error: %s" % _lib.error_name(err.zstdErr))
        if rc == _lib.ERR_SIZE_MISMATCH:
            raise ZstdError("decompression error: decompressed %d bytes; expected %d" % (0, expected))
        if rc == _lib.ERR_NO_MEMORY:
            raise MemoryError()
        if rc != _lib.ERR_NONE:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())
        try:
            ob = out[0]
            produced = ob.segs[0].length
            result = C.string_at(ob.data, produced) if produced else b""
        finally:
            L.zhip_free_outbufs(out, n_out, 1)
        if not allow_extra_data:
            used = L.zhip_find_frame_compressed_size(addr, len(src))
            if 0 <= used < len(src):
                raise ZstdError("compressed input contains %d bytes of unused data, which is disallowed" % (len(src) - used))
        return result

    # ------------------------------------------------------------------------------------------------ batch
    def multi_decompress_to_buffer(self, frames, decompressed_sizes=None, threads=0):
        """Decompress many independent frames on the GPU.

        ``threads`` is accepted for API compatibility (decompressor.c:1484-1494); the work is spread over
        persistent wavefronts, not host threads.
        """
        sizes = None
        if decompressed_sizes is not None:
            smv = memoryview(decompressed_sizes)
            if not smv.c_contiguous:
                raise ValueError("decompressed_sizes buffer should be contiguous and have a single dimension")
            smv = smv.cast("B") if (smv.format != "B" or smv.ndim != 1) else smv
        if isinstance(frames, tuple) or isinstance(frames, (bool, int)) or frames is None:
            raise TypeError("argument must be list or BufferWithSegments")
        views = collect_sources(frames, "argument must be list or BufferWithSegments")
        n = len(views)
        if decompressed_sizes is not None:
            if len(smv) != n * 8:
                raise ValueError("decompressed_sizes size mismatch; expected %d, got %d" % (n * 8, len(smv)))
            sizes = list(struct.unpack("=%dQ" % n, smv.tobytes()))
        if n == 0:
            raise ValueError("no source elements found") if isinstance(frames, list) else ValueError("no source elements found")
        rc, err, out, n_out = self._run(views, sizes, 1 if sizes is not None else 0)
        L = _lib.lib()
        if rc == _lib.ERR_UNKNOWN_SIZE:
            raise ValueError("could not determine decompressed size of item %d" % err.index)
        if rc == _lib.ERR_ZSTD:
            raise ZstdError("error decompressing item %d: %s" % (err.index, _lib.error_name(err.zstdErr)))
        if rc == _lib.ERR_SIZE_MISMATCH:
            raise ZstdError("error decompressing item %d: decompressed %d bytes; expected %d"
                            % (err.index, err.detail[0], err.detail[1]))
        if rc == _lib.ERR_NO_MEMORY:
            raise MemoryError()
        if rc != _lib.ERR_NONE:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())
        buffers = []
        for i in range(n_out):
            ob = out[i]
            buffers.append(BufferWithSegments._from_memory(ob.data, ob.dataSize, C.cast(ob.segs, C.c_void_p).value, ob.nSegs))
        L.zhip_free_outbufs(out, n_out, 0)      # payloads now belong to the BufferWithSegments objects
        return BufferWithSegmentsCollection(*buffers)
