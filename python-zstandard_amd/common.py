"""Shared pieces of the host-side mirror: exception type, dictionary holder, constants."""
import struct

FORMAT_ZSTD1 = 0
FORMAT_ZSTD1_MAGICLESS = 1
MAX_COMPRESSION_LEVEL = 22
DICT_TYPE_AUTO, DICT_TYPE_RAWCONTENT, DICT_TYPE_FULLDICT = 0, 1, 2
DICT_MAGIC = 0xEC30A437


class ZstdError(Exception):
    pass


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
