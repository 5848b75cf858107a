"""ctypes binding of tests/emu/libzhip_emu.so (product kernels compiled for the host wave emulator)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_SO = os.environ.get("ZHIP_EMU_SO") or os.path.join(_DIR, "libzhip_emu.so")      # ZHIP_EMU_SO: e.g. the ASan build (tests/emu/build_asan.sh)


def build():
    subprocess.check_call(["sh", os.path.join(_DIR, "build.sh")])


def build_variant(out_path, defines):
    """the same sources with extra -D flags (experimental kernel variants), e.g. ["-DZP_K3_LONGONE"]"""
    subprocess.check_call(["g++", "-O1", "-g", "-fPIC", "-shared", "-std=c++17", "-I" + _DIR, "-Wno-unused-function", "-Wno-unused-variable"] + list(defines) +
                          ["-o", out_path, os.path.join(_DIR, "zhemu.cpp"), os.path.join(_DIR, "emu_kernels.cpp")])
    return out_path


class Emu:
    def __init__(self, so_path=None):
        path = so_path or EMU_SO
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        if os.environ.get("ZHIP_EMU_PROBES"):                    # stress campaigns: the flat search at four probes per trip (the product's form for chunks up to 32 768 sources)
            self.lib.emu_set_probes(C.c_uint32(int(os.environ["ZHIP_EMU_PROBES"])))

    def set_cparams(self, window_log=0, chain_log=0, hash_log=0, search_log=0, min_match=0, target_length=0, strategy=0, magicless=False):
        """explicit compression parameters / frame format of the following emulated launches (all zero / False = defaults)"""
        self.lib.emu_set_cparams(C.c_uint32(window_log), C.c_uint32(chain_log), C.c_uint32(hash_log), C.c_uint32(search_log),
                                 C.c_uint32(min_match), C.c_uint32(target_length), C.c_int32(strategy), C.c_uint32(1 if magicless else 0))

    def parse_dict(self, dict_bytes):
        """returns (entropy blob or None, content bytes, dictID) using the device dictionary parser under emulation"""
        import struct
        size = self.lib.emu_dict_entropy_size()
        blob = C.create_string_buffer(size)
        d = np.frombuffer(dict_bytes, dtype=np.uint8).copy()
        st = self.lib.emu_parse_dict(d.ctypes.data_as(C.c_void_p), C.c_uint32(len(d)), blob)
        if st:
            raise RuntimeError("dict parse error %d" % st)
        content_off, dict_id, status = struct.unpack_from("<IIi", blob.raw, size - 12)
        huf_count = struct.unpack_from("<I", blob.raw, 256)[0]
        return (blob if huf_count else None), dict_bytes[content_off:], dict_id

    def decompress_batch(self, frames, sizes, n_blocks=2, dict_content=None, dict_id=0, dict_entropy=None):
        n = len(frames)
        src = np.frombuffer(b"".join(frames) + b"\0" * 0, dtype=np.uint8).copy()
        src_segs = np.zeros((n, 2), dtype=np.uint64)
        o = 0
        for i, f in enumerate(frames):
            src_segs[i] = (o, len(f)); o += len(f)
        dst_segs = np.zeros((n, 2), dtype=np.uint64)
        o = 0
        for i, s in enumerate(sizes):
            dst_segs[i] = (o, s); o += s
        dst = np.zeros(max(o, 1), dtype=np.uint8)
        out_sizes = np.zeros(n, dtype=np.uint64)
        status = np.full(n, -1, dtype=np.int32)
        dc = np.frombuffer(dict_content, dtype=np.uint8).copy() if dict_content else None
        self.lib.emu_decompress_batch(
            src.ctypes.data_as(C.c_void_p), src_segs.ctypes.data_as(C.c_void_p), C.c_uint32(n),
            dst.ctypes.data_as(C.c_void_p), dst_segs.ctypes.data_as(C.c_void_p),
            out_sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
            dc.ctypes.data_as(C.c_void_p) if dc is not None else None, C.c_uint32(len(dc) if dc is not None else 0),
            C.c_uint32(dict_id), dict_entropy, C.c_uint32(n_blocks))
        outs = []
        for i in range(n):
            a = int(dst_segs[i][0])
            outs.append(dst[a:a + int(out_sizes[i])].tobytes())
        return outs, status.tolist()


def _emu_compress_batch(self, raws, level=3, flags=5, n_blocks=2, pipeline=False, chunk=0, dict_data=None):
    n = len(raws)
    if dict_data:
        d = np.frombuffer(dict_data, dtype=np.uint8).copy()
        st = self.lib.emu_set_cdict(d.ctypes.data_as(C.c_void_p), C.c_uint32(len(d)), C.c_int(level))
        if st:
            raise RuntimeError("cdict error %d" % st)
    else:
        self.lib.emu_set_cdict(None, C.c_uint32(0), C.c_int(level))
    src = np.frombuffer(b"".join(raws) + b"\0" * 16, dtype=np.uint8).copy()
    src_segs = np.zeros((n, 2), dtype=np.uint64)
    dst_segs = np.zeros((n, 2), dtype=np.uint64)
    o = d = 0
    for i, r in enumerate(raws):
        src_segs[i] = (o, len(r)); o += len(r)
        b = len(r) + (len(r) >> 8) + (((128 << 10) - len(r)) >> 11 if len(r) < (128 << 10) else 0)
        dst_segs[i] = (d, b); d += b
    dst = np.zeros(max(d, 1) + 16, dtype=np.uint8)
    out_sizes = np.zeros(n, dtype=np.uint64)
    status = np.full(n, -1, dtype=np.int32)
    args = [src.ctypes.data_as(C.c_void_p), src_segs.ctypes.data_as(C.c_void_p), C.c_uint32(n),
            dst.ctypes.data_as(C.c_void_p), dst_segs.ctypes.data_as(C.c_void_p),
            out_sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
            C.c_int(level), C.c_uint32(flags), C.c_uint32(n_blocks)]
    if pipeline:
        self.lib.emu_compress_pipeline(*args, C.c_uint32(chunk))
    else:
        self.lib.emu_compress_batch(*args)
    outs = [dst[int(dst_segs[i][0]):int(dst_segs[i][0]) + int(out_sizes[i])].tobytes() for i in range(n)]
    return outs, status.tolist()


Emu.compress_batch = _emu_compress_batch


def _emu_decompress_pipeline(self, frames, sizes, n_blocks=2, chunk=0):
    n = len(frames)
    src = np.frombuffer(b"".join(frames) + b"\0" * 16, dtype=np.uint8).copy()
    src_segs = np.zeros((n, 2), dtype=np.uint64)
    o = 0
    for i, f in enumerate(frames):
        src_segs[i] = (o, len(f)); o += len(f)
    dst_segs = np.zeros((n, 2), dtype=np.uint64)
    o = 0
    for i, s in enumerate(sizes):
        dst_segs[i] = (o, s); o += s
    dst = np.zeros(max(o, 1) + 16, dtype=np.uint8)
    out_sizes = np.zeros(n, dtype=np.uint64)
    status = np.full(n, -1, dtype=np.int32)
    nfb = self.lib.emu_decompress_pipeline(src.ctypes.data_as(C.c_void_p), src_segs.ctypes.data_as(C.c_void_p), C.c_uint32(n),
                                           dst.ctypes.data_as(C.c_void_p), dst_segs.ctypes.data_as(C.c_void_p),
                                           out_sizes.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                                           C.c_uint32(n_blocks), C.c_uint32(chunk))
    outs = [dst[int(dst_segs[i][0]):int(dst_segs[i][0]) + int(out_sizes[i])].tobytes() for i in range(n)]
    return outs, status.tolist(), nfb


Emu.decompress_pipeline = _emu_decompress_pipeline


def _emu_set_ddict(self, dict_data, raw_content=False):
    """decompression dictionary of the pipeline harness (None: no dictionary); returns the zstd error code of its digestion, 0 = fine"""
    if not dict_data:
        return self.lib.emu_set_ddict(None, C.c_uint32(0), C.c_int(0))
    d = np.frombuffer(dict_data, dtype=np.uint8).copy()
    return self.lib.emu_set_ddict(d.ctypes.data_as(C.c_void_p), C.c_uint32(len(d)), C.c_int(1 if raw_content else 0))


Emu.set_ddict = _emu_set_ddict


def _emu_set_blocks(self, per_frame):
    """several-block mode of the decode pipeline harness: item slots per frame (0 = off: frames of several blocks go to the generic kernel)"""
    self.lib.emu_set_blocks(C.c_uint32(per_frame))


Emu.set_blocks = _emu_set_blocks


def _emu_set_mb_compress(self, v):
    """sources of several blocks in the flat match kernel of the compress pipeline harness: 0 off (the generic kernel searches them), 1 on,
    > 1 on with that many block slots per frame"""
    self.lib.emu_set_mb_compress(C.c_uint32(v))


def _emu_set_mb_hint(self, v):
    """size the several-block compress arenas from this hint instead of the batch's largest source (0 = the largest source): what a device-API
    caller's stale zhip_ctx_set_size_hint does"""
    self.lib.emu_set_mb_hint(C.c_uint64(v))


def _emu_set_dict_slot_max(self, v):
    """dictionary batches: the match kernels take sources up to v bytes (their slots' size under a caller's size hint), larger ones go to the generic kernel; 0 = the attach cutoff"""
    self.lib.emu_set_dict_slot_max(C.c_uint32(v))


def _emu_stat(self, i):
    self.lib.emu_stat.restype = C.c_long
    return int(self.lib.emu_stat(C.c_int(i)))


Emu.set_mb_compress = _emu_set_mb_compress
Emu.set_mb_hint = _emu_set_mb_hint
Emu.set_dict_slot_max = _emu_set_dict_slot_max
Emu.stat = _emu_stat


def _emu_set_k0(self, v):
    """1 (default, as the product): K0 -- a lane per frame walks K1's serial descriptions -- runs before K1; 0: K1 parses everything itself"""
    self.lib.emu_set_k0(C.c_uint32(v))


Emu.set_k0 = _emu_set_k0


def _emu_set_dict_epochs(self, v):
    """1: the flat dictionary search's tables persist between calls and carry launch numbers in their cells (the product's way); 0 (default): the kernel's waves zero them"""
    self.lib.emu_set_dict_epochs(C.c_uint32(v))


Emu.set_dict_epochs = _emu_set_dict_epochs


def _emu_set_check_later(self, v):
    """1 (default, as the product): content checksums are verified by KX after K3, a lane per frame; 0: by K1 / K3 on one lane at the frame's end"""
    self.lib.emu_set_check_later(C.c_uint32(v))


Emu.set_check_later = _emu_set_check_later
