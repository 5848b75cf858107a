"""Shared pieces of the host-side mirror: exception type, dictionary holder, constants."""
import struct

FORMAT_ZSTD1 = 0
FORMAT_ZSTD1_MAGICLESS = 1
MAX_COMPRESSION_LEVEL = 22
DICT_TYPE_AUTO, DICT_TYPE_RAWCONTENT, DICT_TYPE_FULLDICT = 0, 1, 2
DICT_MAGIC = 0xEC30A437


# frame / format constants the reference exports (c-ext/constants.c:54-100); values of zstd.h
CONTENTSIZE_UNKNOWN = 2**64 - 1
CONTENTSIZE_ERROR = 2**64 - 2
MAGIC_NUMBER = 0xFD2FB528
BLOCKSIZE_MAX = 1 << 17
COMPRESSION_RECOMMENDED_INPUT_SIZE = 1 << 17
COMPRESSION_RECOMMENDED_OUTPUT_SIZE = (1 << 17) + 512 + 3 + 4           # ZSTD_compressBound(block) + block header + checksum
DECOMPRESSION_RECOMMENDED_INPUT_SIZE = (1 << 17) + 3
DECOMPRESSION_RECOMMENDED_OUTPUT_SIZE = 1 << 17
WINDOWLOG_MIN, WINDOWLOG_MAX = 10, 31


class ZstdError(Exception):
    pass


class FrameParameters:
    """What ZSTD_getFrameHeader reports (c-ext/frameparams.c:13-75)."""

    __slots__ = ("content_size", "window_size", "dict_id", "has_checksum")

    def __init__(self, content_size, window_size, dict_id, has_checksum):
        self.content_size, self.window_size, self.dict_id, self.has_checksum = content_size, window_size, dict_id, has_checksum


def get_frame_parameters(data, format=FORMAT_ZSTD1):
    """Parse a frame header (RFC 8878 3.1.1.1; ZSTD_getFrameHeader_advanced zstd.c:43668) on the host."""
    mv = memoryview(data).cast("B")
    if format != FORMAT_ZSTD1:
        raise ZstdError("cannot get frame parameters: only FORMAT_ZSTD1 is supported by the HIP backend")
    if len(mv) < 5:
        raise ZstdError("not enough data for frame parameters; need %d bytes" % 5)
    if struct.unpack_from("<I", mv)[0] != MAGIC_NUMBER:
        raise ZstdError("cannot get frame parameters: Unknown frame descriptor")
    fhd = mv[4]
    fcs_code, single, checksum, did_code = fhd >> 6, (fhd >> 5) & 1, (fhd >> 2) & 1, fhd & 3
    if fhd & 8:
        raise ZstdError("cannot get frame parameters: Unsupported frame parameter")
    did_size = (0, 1, 2, 4)[did_code]
    fcs_size = (1 if single else 0, 2, 4, 8)[fcs_code]
    need = 5 + (0 if single else 1) + did_size + fcs_size
    if len(mv) < need:
        raise ZstdError("not enough data for frame parameters; need %d bytes" % need)
    pos = 5
    window = 0
    if not single:
        b = mv[pos]; pos += 1
        wlog = 10 + (b >> 3)
        if wlog > WINDOWLOG_MAX:
            raise ZstdError("cannot get frame parameters: Frame requires too much memory for decoding")
        window = (1 << wlog) + ((1 << wlog) >> 3) * (b & 7)
    dict_id = int.from_bytes(mv[pos:pos + did_size], "little") if did_size else 0
    pos += did_size
    if fcs_size == 0:
        content = CONTENTSIZE_UNKNOWN
    else:
        content = int.from_bytes(mv[pos:pos + fcs_size], "little") + (256 if fcs_size == 2 else 0)
    if single:
        window = content
    return FrameParameters(content, window, dict_id, bool(checksum))


class ZstdCompressionDict:
    """Raw bytes of a dictionary (consumer side of c-ext/compressiondict.c:164-348).

    Training (ZDICT_*, compressiondict.c:13-146) is a one-off CPU job outside the hot path; use the reference for it.
    """

    def __init__(self, data, dict_type=DICT_TYPE_AUTO):
        if dict_type not in (DICT_TYPE_AUTO, DICT_TYPE_RAWCONTENT, DICT_TYPE_FULLDICT):
            raise ValueError("invalid dictionary load mode: %d; must use DICT_TYPE_* constants" % dict_type)
        self._data = bytes(data)
        self._dict_type = dict_type

    def __len__(self):
        return len(self._data)

    def as_bytes(self):
        return self._data

    def dict_id(self):
        d = self._data
        if self._dict_type != DICT_TYPE_RAWCONTENT and len(d) >= 8 and struct.unpack_from("<I", d)[0] == DICT_MAGIC:
            return struct.unpack_from("<I", d, 4)[0]
        return 0

    def precompute_compress(self, level=0, compression_params=None):
        # CDict tables are built on the device when the dictionary is attached to a compressor context
        return None


def collect_sources(data, type_error_message="argument must be list of BufferWithSegments"):
    """Flatten the three accepted input shapes into [(memoryview, address-owner)] like compressor.c:1369-1466."""
    from .buffers import BufferWithSegments, BufferWithSegmentsCollection

    items = []
    if isinstance(data, BufferWithSegments):
        for i in range(len(data)):
            off, ln = data._segment_bounds(i)
            items.append(data._data[off:off + ln])
    elif isinstance(data, BufferWithSegmentsCollection):
        for b in data._buffers:
            for i in range(len(b)):
                off, ln = b._segment_bounds(i)
                items.append(b._data[off:off + ln])
    elif isinstance(data, list):
        for i, o in enumerate(data):
            try:
                mv = memoryview(o)
            except TypeError:
                raise TypeError("item %d not a bytes like object" % i)
            if not mv.c_contiguous:
                raise TypeError("item %d not a bytes like object" % i)
            items.append(mv.cast("B") if (mv.format != "B" or mv.ndim != 1) else mv)
    else:
        raise TypeError(type_error_message)
    return items
