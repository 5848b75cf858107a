"""Device-resident batch API: frames already in HBM in, frames in HBM out (torch tensors only carry the memory).

This is the same hot path as ``multi_*_to_buffer`` without the PCIe hops, and it is what shards across GPUs:
each rank owns a contiguous range of frames (the reference's partition rule, c-ext/compressor.c:1127-1216) and
results stay resident per GPU.
"""
import ctypes as C

import torch

from . import _lib
from .backend_hip import ZstdError


class DeviceBatchContext:
    """Owns the native context (scratch, dictionary tables, kernel timers) for one GPU / one stream."""

    def __init__(self, dict_data=None, level=3, write_checksum=False, write_content_size=True, write_dict_id=True, dict_type=0,
                 format=0, max_window_size=0, **cparams):
        """dict_data: a ZstdCompressionDict (or bytes) used by both directions; level / write_* / cparams (window_log, hash_log,
        chain_log, min_match, target_length, strategy): what ZstdCompressor takes; format / max_window_size: what ZstdDecompressor takes."""
        self.L = _lib.lib()
        self.ctx = self.L.zhip_ctx_create()
        if not self.ctx:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())
        raw = None if dict_data is None else (dict_data.as_bytes() if hasattr(dict_data, "as_bytes") else bytes(dict_data))
        self._dict_buf = C.create_string_buffer(raw, len(raw)) if raw else None          # kept alive: the library fingerprints it per call
        rc = self.L.zhip_ctx_set_dformat(self.ctx, format, max_window_size)
        if rc:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())
        if raw:
            rc = self.L.zhip_ctx_set_ddict(self.ctx, C.cast(self._dict_buf, C.c_void_p), len(raw), dict_type)
            if rc:
                raise ZstdError("could not load dictionary: %s" % (_lib.error_name(-rc) if rc < 0 else _lib.last_error()))
        p = _lib.CParams()
        p.level, p.contentSizeFlag, p.checksumFlag, p.dictIDFlag = level, int(write_content_size), int(write_checksum), int(write_dict_id)
        p.dictType, p.format = dict_type, format
        names = {"window_log": "windowLog", "chain_log": "chainLog", "hash_log": "hashLog", "search_log": "searchLog",
                 "min_match": "minMatch", "target_length": "targetLength", "strategy": "strategy"}
        for k, v in cparams.items():
            setattr(p.cp, names[k], v)
        if raw:
            p.dict, p.dictSize = C.cast(self._dict_buf, C.c_void_p), len(raw)
        # the compression side is set up by the first compress() call: digesting the dictionary for compression (tagged tables, entropy
        # encoding tables) is work a decode-only context never needs, and a dictionary the compressor refuses must not stop a decoder
        self._cparams, self._cparams_set = p, False

    def _ensure_cparams(self):
        if not self._cparams_set:
            rc = self.L.zhip_ctx_set_cparams(self.ctx, C.byref(self._cparams))
            if rc:
                raise ZstdError("could not set compression parameters: %s" % (_lib.error_name(-rc) if rc < 0 else _lib.last_error()))
            self._cparams_set = True

    def set_size_hint(self, max_item_bytes):
        """largest uncompressed item of the coming calls (0 = unknown): with items above 128 KiB (frames of several blocks) decompression
        runs the phase-split kernels in their several-block mode and compression of large batches gives those sources to the flat match
        kernel; untold, such items are served one wave each by a token grid of the generic kernels (correct, slow)"""
        self.L.zhip_ctx_set_size_hint(self.ctx, int(max_item_bytes))

    def close(self):
        if self.ctx:
            self.L.zhip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(t, dtype):
        assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, "expected a contiguous CUDA tensor of %s" % dtype

    def decompress(self, src, src_segs, dst, dst_segs, out_sizes, status, stream=None):
        """Asynchronous. src/dst: uint8 arenas; *_segs: int64 [n,2] (offset,length|capacity); out_sizes int64[n]; status int32[n]."""
        self._check(src, torch.uint8); self._check(dst, torch.uint8)
        self._check(src_segs, torch.int64); self._check(dst_segs, torch.int64)
        self._check(out_sizes, torch.int64); self._check(status, torch.int32)
        n = src_segs.shape[0]
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = self.L.zhip_decompress_batch_device(self.ctx, src.data_ptr(), src_segs.data_ptr(), n, dst.data_ptr(),
                                                 dst_segs.data_ptr(), out_sizes.data_ptr(), status.data_ptr(),
                                                 s.cuda_stream)
        if rc:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())

    def compress(self, src, src_segs, dst, dst_segs, out_sizes, status, stream=None):
        self._ensure_cparams()
        self._check(src, torch.uint8); self._check(dst, torch.uint8)
        n = src_segs.shape[0]
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = self.L.zhip_compress_batch_device(self.ctx, src.data_ptr(), src_segs.data_ptr(), n, dst.data_ptr(),
                                               dst_segs.data_ptr(), out_sizes.data_ptr(), status.data_ptr(),
                                               s.cuda_stream)
        if rc:
            raise ZstdError("HIP backend failure: %s" % _lib.last_error())

    def kernel_time(self, direction):
        """(average ms per launch, launches) of the dominant kernel since the last call, from HIP events on the launch stream."""
        ms, n = C.c_double(0), C.c_uint64(0)
        self.L.zhip_ctx_kernel_time(self.ctx, direction, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def table_pick(self):
        """(candidate times in ms -- 0 = not tried --, index kept) of the compress direction's table placement pick; zeros while none has happened"""
        ms = (C.c_float * 3)()
        kept = self.L.zhip_ctx_table_pick(self.ctx, ms)
        return [float(x) for x in ms], int(kept)

    def kernel_name(self, direction):
        return self.L.zhip_kernel_name(direction).decode()
