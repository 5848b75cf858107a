"""Hand-made zstd frames for cases no encoder emits (test infrastructure).

`sequences_block()` writes a compressed block whose sequences use the PREDEFINED FSE tables (RFC 8878 3.1.1.3.2.2), from explicit
(literal length, match length, offset VALUE) triples -- the offset value is the number in the stream (1..3 = repeat codes, n + 3 = a
new offset n), so a test can ask for "repeat offset 1 minus one" where that is zero, which libzstd 1.5.7 refuses
(zstd/zstd.c:46941 "0 is not valid: input corrupted => force offset to -1").  The tables are built as the format says (the decoder's
spread and state assignment); encoding walks the sequences backwards choosing, for every symbol, the one state whose range holds the
state that follows.  Values are read from oracle/zo_common's tables through the oracle library so that nothing is typed twice."""
import ctypes as C

from tests import reflib

_LL_BASE = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096,
            8192, 16384, 32768, 65536]
_LL_BITS = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
_ML_BASE = list(range(3, 35)) + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539]
_ML_BITS = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]


def _defnorm(name, n):
    lib = C.CDLL(reflib.ORACLE_SO)
    return list((C.c_int16 * n).in_dll(lib, name))


def _decode_table(norm, log):
    """[(symbol, nbBits, base)] per state, as the format's decoder builds it."""
    size = 1 << log
    sym = [0] * size
    high = size - 1
    for s, c in enumerate(norm):
        if c == -1:
            sym[high] = s; high -= 1
    step, mask, pos = (size >> 1) + (size >> 3) + 3, size - 1, 0
    for s, c in enumerate(norm):
        for _ in range(max(c, 0)):
            sym[pos] = s
            pos = (pos + step) & mask
            while pos > high: pos = (pos + step) & mask
    nxt = [1 if c == -1 else c for c in norm]
    cells = []
    for x in range(size):
        s = sym[x]; n = nxt[s]; nxt[s] += 1
        nb = log - (n.bit_length() - 1)
        cells.append((s, nb, (n << nb) - size))
    return cells


def _code(value, base):
    c = 0
    for i, b in enumerate(base):
        if b <= value: c = i
    return c


def sequences_block(literals, seqs, last=True, of_rle=False):
    """One compressed block: raw literals + `seqs` = [(ll, ml, offset_value)] on predefined tables. Returns block header + content.
    of_rle: the offset codes in RLE mode (one code for every sequence, any of 0..31 -- the predefined table ends at 28)."""
    assert len(literals) < 32 and 0 < len(seqs) < 128
    tabs = [(_decode_table(_defnorm("zo_ll_defnorm", 36), 6), 6), (_decode_table(_defnorm("zo_of_defnorm", 29), 5), 5),
            (_decode_table(_defnorm("zo_ml_defnorm", 53), 6), 6)]
    if of_rle:
        oc = seqs[0][2].bit_length() - 1
        assert all(q[2].bit_length() - 1 == oc for q in seqs)
        tabs[1] = ([(oc, 0, 0)], 0)
    codes = []
    for ll, ml, ofv in seqs:
        lc, mc, oc = _code(ll, _LL_BASE), _code(ml, _ML_BASE), ofv.bit_length() - 1
        codes.append(((lc, ll - _LL_BASE[lc], _LL_BITS[lc]), (oc, ofv - (1 << oc), oc), (mc, ml - _ML_BASE[mc], _ML_BITS[mc])))
    # states, last sequence first: any state of the symbol for the last one, then the state whose range holds its successor
    states = [None] * len(seqs)
    for n in range(len(seqs) - 1, -1, -1):
        st = []
        for k in range(3):
            cells = tabs[k][0]; want = codes[n][k][0]
            if n == len(seqs) - 1: x = next(i for i, c in enumerate(cells) if c[0] == want); bits = None
            else:
                succ = states[n + 1][k][0]
                x = next(i for i, c in enumerate(cells) if c[0] == want and c[2] <= succ < c[2] + (1 << c[1]))
                bits = (succ - cells[x][2], cells[x][1])
            st.append((x, bits))
        states[n] = st
    # the fields in the order the decoder reads them (from the top of the stream down)
    reads = [(states[0][0][0], 6), (states[0][1][0], tabs[1][1]), (states[0][2][0], 6)]
    for n in range(len(seqs)):
        (_, lx, lb), (_, ox, ob), (_, mx, mb) = codes[n]
        reads += [(ox, ob), (mx, mb), (lx, lb)]
        if n + 1 < len(seqs): reads += [states[n][0][1], states[n][2][1], states[n][1][1]]      # LL, ML, OF state bits
    acc = 1
    for v, nb in reads: acc = (acc << nb) | v
    stream = acc.to_bytes((acc.bit_length() + 7) // 8, "little")
    content = bytes([len(literals) << 3]) + bytes(literals) + (bytes([len(seqs), 0x10, tabs[1][0][0][0]]) if of_rle else bytes([len(seqs), 0])) + stream
    bh = (1 if last else 0) | (2 << 1) | (len(content) << 3)
    return bh.to_bytes(3, "little") + content


def raw_block(data, last=False):
    bh = (1 if last else 0) | (len(data) << 3)
    return bh.to_bytes(3, "little") + bytes(data)


def frame(blocks, content_size):
    """magic + single-segment header with a one-byte content size + blocks"""
    assert content_size < 256
    return b"\x28\xb5\x2f\xfd" + bytes([0x20, content_size]) + b"".join(blocks)


def _hdr(fcs=None, window_log=None):
    """frame header: single-segment with the content size, or a window descriptor with / without it"""
    magic = b"\x28\xb5\x2f\xfd"
    if window_log is None:
        assert fcs is not None and fcs < 256
        return magic + bytes([0x20, fcs])
    wd = bytes([(window_log - 10) << 3])
    if fcs is None: return magic + b"\x00" + wd
    return magic + b"\x80" + wd + fcs.to_bytes(4, "little")


def rle_block(byte, n, last=True):
    return ((1 if last else 0) | (1 << 1) | (n << 3)).to_bytes(3, "little") + bytes([byte])


def edge_frames():
    """(name, frame, declared size, accepted by libzstd 1.5.7 as the reference drives it) -- what an encoder never writes but a decoder
    must answer like libzstd: zstd.c:46941 (repeat offset 1 minus one = 0), :47714 / :44239-44246 (block sizes against the frame's
    maximum in the one-pass decoder) and ZSTD_decompressContinue's stricter check when the content size is not in the header."""
    lit = b"abcdefgh"
    pat = bytes(range(256)) * 8
    out = [
        ("rep0 minus one is zero", frame([sequences_block(lit, [(8, 30, 4), (0, 3, 3)])], 41), 41, False),
        ("rep0 minus one is one", frame([sequences_block(lit, [(8, 30, 5), (0, 3, 3)])], 41), 41, True),
        ("repeat codes with empty literal runs", frame([sequences_block(lit, [(8, 30, 4), (0, 3, 2), (0, 4, 3)])], 45), 45, True),
        ("second block: rep0 minus one is zero", frame([raw_block(lit), sequences_block(b"", [(0, 3, 3)])], 11), 11, False),
        ("second block inherits history", frame([raw_block(lit), sequences_block(b"xy", [(1, 4, 2), (0, 3, 2), (1, 3, 1)])], 20), 20, True),
        ("compressed block above the frame's maximum", frame([sequences_block(lit, [(8, 3, 4)])], 11), 11, False),
        ("raw block above the window, size known", _hdr(2000, 10) + raw_block(pat[:2000], True), 2000, True),
        ("raw block above the window, size unknown", _hdr(None, 10) + raw_block(pat[:2000], True), 2000, False),
        ("RLE block above the window, size known", _hdr(2000, 10) + rle_block(7, 2000), 2000, True),
        ("RLE block above the window, size unknown", _hdr(None, 10) + rle_block(7, 2000), 2000, False),
        ("RLE block above 128 KiB, size known", _hdr(200000, 20) + rle_block(9, 200000), 200000, True),
        ("RLE block above 128 KiB, size unknown", _hdr(None, 20) + rle_block(9, 200000), 200000, False),
        ("raw block above 128 KiB, size known", _hdr(200000, 20) + raw_block((pat * 98)[:200000], True), 200000, True),
        ("raw blocks then RLE above the window", _hdr(5000, 10) + raw_block(pat[:1500]) + raw_block(pat[:1500]) + rle_block(3, 2000), 5000, True),
        # offsets the decode kernels' packed sequences cannot hold (2^29 and up): never valid in a frame this small
        ("offset codes in RLE mode", frame([sequences_block(lit, [(4, 5, 4), (4, 3, 6)], of_rle=True)], 16), 16, True),
        ("an offset of 768 MiB", frame([sequences_block(lit, [(8, 30, 0x30000003)], of_rle=True)], 38), 38, False),
        ("an offset of 3 GiB in a second block", frame([raw_block(lit), sequences_block(b"xy", [(1, 4, 0x80000007), (1, 3, 0xC0000003)], of_rle=True)], 17), 17, False),
        ("offsets around the packed form's limit", frame([sequences_block(lit, [(4, 3, 0x1E000002), (4, 3, 0x1E000003)], of_rle=True)], 14), 14, False),
    ]
    return out


def skippable_frames():
    """(name, item bytes, declared size, accepted) -- skippable frames (RFC 8878 3.1.2) as batch items: ZSTD_decompressStream passes over one
    and stops at the frame boundary with nothing produced (zstd.c:43706-43715, :44731), so the reference returns an empty segment;
    what follows the first frame of an item is never looked at (c-ext/decompressor.c:1150-1163)."""
    skip = b"\x50\x2a\x4d\x18" + (5).to_bytes(4, "little") + b"hello"
    whole = frame([sequences_block(b"abcdefgh", [(8, 30, 5), (0, 3, 3)])], 41)
    return [
        ("skippable frame", skip, 0, True),
        ("skippable frame, last magic, no payload", b"\x5f\x2a\x4d\x18" + bytes(4), 0, True),
        ("skippable frame, then a frame", skip + whole, 0, True),
        ("a frame, then a skippable frame", whole + skip, 41, True),
        ("a frame, then bytes that are no frame", whole + b"garbage!", 41, True),
        ("skippable frame, payload cut short", skip[:-2], 0, False),
        ("skippable frame, header cut short", skip[:6], 0, False),
        ("one past the skippable magics", b"\x60\x2a\x4d\x18" + skip[4:], 0, False),
    ]


def encoding_variants():
    """(name, frame, declared size, accepted) -- legal but non-minimal encodings and small illegal ones around the section headers: literal
    sizes in the longer header forms, the sequence count in its 2- and 3-byte forms, RLE-mode tables with symbols beyond the alphabets,
    reserved mode bits, degenerate block sizes, a treeless literals section with nothing to inherit. All agreed with libzstd when they were
    written (round 3); they stay as a fence."""
    magic = b"\x28\xb5\x2f\xfd"

    def blk(content, last=True): return ((1 if last else 0) | (2 << 1) | (len(content) << 3)).to_bytes(3, "little") + content

    def frm(blocks): return magic + bytes([0x00, (17 - 10) << 3]) + b"".join(blocks)
    lits = b"abcdefgh"
    seqpart = sequences_block(lits, [(8, 30, 5), (0, 3, 3)])[3:][1 + 8:]              # count, modes, stream
    rl_seq = sequences_block(b"aaaaaaaa", [(8, 30, 5), (0, 3, 3)])[3:][1 + 8:]

    def seq_rle(ll, of, ml, stream): return bytes([1, (1 << 6) | (1 << 4) | (1 << 2), ll, of, ml]) + stream
    h1 = bytes([8 << 3])
    out = [
        ("raw literals, 2-byte header", frm([blk(((8 << 4) | (1 << 2)).to_bytes(2, "little") + lits + seqpart)]), 41, True),
        ("raw literals, 3-byte header", frm([blk(((8 << 4) | (3 << 2)).to_bytes(3, "little") + lits + seqpart)]), 41, True),
        ("RLE literals, 1-byte header", frm([blk(bytes([(8 << 3) | 1]) + b"a" + rl_seq)]), 41, True),
        ("RLE literals, 2-byte header", frm([blk(((8 << 4) | (1 << 2) | 1).to_bytes(2, "little") + b"a" + rl_seq)]), 41, True),
        ("RLE literals, 3-byte header", frm([blk(((8 << 4) | (3 << 2) | 1).to_bytes(3, "little") + b"a" + rl_seq)]), 41, True),
        ("sequence count in the 2-byte form", frm([blk(h1 + lits + bytes([0x80, 2]) + seqpart[1:])]), 41, True),
        ("sequence count in the 3-byte form, far too many", frm([blk(h1 + lits + bytes([0xFF, 0x02, 0x81]) + seqpart[1:])]), 41, False),
        ("no sequences, count in the 2-byte form", frm([blk(h1 + lits + bytes([0x80, 0]))]), 8, True),
        ("no sequences but bytes after the count", frm([blk(h1 + lits + bytes([0, 0]))]), 8, False),
        ("RLE-mode tables", frm([blk(h1 + lits + seq_rle(8, 2, 0, bytes([0b100])))]), 11, True),
        ("RLE-mode literal-length symbol 36", frm([blk(h1 + lits + seq_rle(36, 2, 0, bytes([1])))]), 11, False),
        ("RLE-mode offset symbol 32", frm([blk(h1 + lits + seq_rle(8, 32, 0, bytes([1])))]), 11, False),
        ("RLE-mode match-length symbol 53", frm([blk(h1 + lits + seq_rle(8, 2, 53, bytes([1])))]), 11, False),
        ("reserved mode bits set", frm([blk(h1 + lits + bytes([2, 1]) + seqpart[2:])]), 41, False),
        ("compressed block of one byte", _hdr(0, 17) + blk(b"\x00"), 0, False),                       # (content size 0 in the header: a batch item needs one)
        ("compressed block without literals or sequences", _hdr(0, 17) + blk(b"\x00\x00"), 0, True),
        ("treeless literals with nothing to inherit", frm([blk(bytes([0x83, 0x40, 0x01]) + b"\x01\x02\x03\x04\x05" + bytes([0]))]), 8, False),
    ]
    return out
