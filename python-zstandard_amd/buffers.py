"""Buffer types of the batch APIs, mirroring c-ext/bufferutil.c (types declared at c-ext/python-zstandard.h:307-368).

A ``BufferWithSegments`` is one contiguous payload plus an array of ``BufferSegment`` records
(2 x native-endian u64: offset, length). Error messages are the reference's (pinned by tests/test_buffer_util.py).
"""
import ctypes
import struct

_SEG = struct.Struct("=QQ")


def _as_memoryview(obj, what):
    try:
        mv = memoryview(obj)
    except TypeError:
        raise TypeError("%s must be a bytes-like object" % what)
    if not mv.c_contiguous:
        raise ValueError("%s must be C contiguous" % what)
    return mv.cast("B") if mv.format != "B" or mv.ndim != 1 else mv


class BufferSegment:
    """One (offset, length) slice of a parent BufferWithSegments (bufferutil.c:307-360)."""

    __slots__ = ("_parent", "_mv", "offset")

    def __init__(self, *a, **k):
        raise TypeError("cannot create 'BufferSegment' instances directly")

    @classmethod
    def _make(cls, parent, mv, offset):
        self = object.__new__(cls)
        self._parent, self._mv, self.offset = parent, mv, offset
        return self

    def __len__(self):
        return len(self._mv)

    def tobytes(self):
        return self._mv.tobytes()

    def __bytes__(self):
        return self._mv.tobytes()

    def memoryview(self):
        return self._mv


class BufferSegments:
    """The raw segment table of a BufferWithSegments (bufferutil.c:271-305)."""

    __slots__ = ("_parent", "_mv")

    def __init__(self, *a, **k):
        raise TypeError("cannot create 'BufferSegments' instances directly")

    @classmethod
    def _make(cls, parent, mv):
        self = object.__new__(cls)
        self._parent, self._mv = parent, mv
        return self

    def tobytes(self):
        return self._mv.tobytes()

    def memoryview(self):
        return self._mv


class _Owned:
    """malloc()ed memory handed over by the C ABI (zhip_outbuf); freed with free() like useFree=1 does."""

    _libc = ctypes.CDLL(None)
    _libc.free.argtypes = [ctypes.c_void_p]
    _libc.free.restype = None

    def __init__(self, addr, size):
        self.addr, self.size = addr, size
        self.array = (ctypes.c_ubyte * max(size, 1)).from_address(addr) if addr else None

    def view(self):
        return memoryview(self.array).cast("B")[: self.size] if self.array is not None else memoryview(b"")

    def __del__(self):
        if getattr(self, "addr", None):
            self.array = None
            self._libc.free(self.addr)
            self.addr = None


class BufferWithSegments:
    def __init__(self, data, segments):
        self._data = _as_memoryview(data, "data")
        self._segs = _as_memoryview(segments, "segments")
        self._owners = (data, segments)
        if len(self._segs) % _SEG.size:
            raise ValueError("segments array size is not a multiple of %d" % _SEG.size)
        self._n = len(self._segs) // _SEG.size
        size = len(self._data)
        q = self._segs.cast("Q") if self._n else ()
        for i in range(self._n):
            if q[2 * i] + q[2 * i + 1] > size:
                raise ValueError("offset within segments array references memory outside buffer")
        self._q = q

    @classmethod
    def _from_memory(cls, data_addr, data_size, segs_addr, n_segs):
        """BufferWithSegments_FromMemory (bufferutil.c:107-148): takes ownership of two malloc()ed blocks."""
        self = object.__new__(cls)
        d, s = _Owned(data_addr, data_size), _Owned(segs_addr, n_segs * _SEG.size)
        self._owners = (d, s)
        self._data, self._segs, self._n = d.view(), s.view(), n_segs
        self._q = self._segs.cast("Q") if n_segs else ()
        for i in range(n_segs):
            if self._q[2 * i] + self._q[2 * i + 1] > data_size:
                raise ValueError("offset in segments overflows buffer size")
        return self

    @property
    def size(self):
        return len(self._data)

    def __len__(self):
        return self._n

    def _segment_bounds(self, i):
        return self._q[2 * i], self._q[2 * i + 1]

    def __getitem__(self, i):
        if not isinstance(i, int):
            raise TypeError("indices must be integers")
        if i < 0:
            raise IndexError("offset must be non-negative")
        if i >= self._n:
            raise IndexError("offset must be less than %d" % self._n)
        off, ln = self._segment_bounds(i)
        return BufferSegment._make(self, self._data[off:off + ln], off)

    def segments(self):
        return BufferSegments._make(self, self._segs)

    def tobytes(self):
        return self._data.tobytes()

    def memoryview(self):
        return self._data

    # address of the payload for zero-copy hand-off to the C ABI
    def _address(self):
        return _addr_of(self._data)


def _addr_of(mv):
    """address of a (possibly read-only) contiguous memoryview without copying"""
    import numpy as np
    return np.frombuffer(mv, dtype=np.uint8).ctypes.data if len(mv) else 0


class BufferWithSegmentsCollection:
    """Virtual concatenation of several BufferWithSegments (bufferutil.c:362-528)."""

    def __init__(self, *args):
        if not args:
            raise ValueError("must pass at least 1 argument")
        for a in args:
            if not isinstance(a, BufferWithSegments):
                raise TypeError("arguments must be BufferWithSegments instances")
            if len(a) == 0 or a.size == 0:
                raise ValueError("ZstdBufferWithSegments cannot be empty")
        self._buffers = list(args)
        self._first = []
        total = 0
        for b in self._buffers:
            total += len(b)
            self._first.append(total)

    def __len__(self):
        return self._first[-1]

    def size(self):
        return sum(ln for b in self._buffers for ln in (b._segment_bounds(i)[1] for i in range(len(b))))

    def __getitem__(self, i):
        if not isinstance(i, int):
            raise TypeError("indices must be integers")
        if i < 0:
            raise IndexError("offset must be non-negative")
        if i >= len(self):
            raise IndexError("offset must be less than %d" % len(self))
        prev = 0
        for b, first in zip(self._buffers, self._first):
            if i < first:
                return b[i - prev]
            prev = first
        raise RuntimeError("error resolving segment; this should not happen")
