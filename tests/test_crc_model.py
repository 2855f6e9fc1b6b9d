"""The algebra of the CRC-32 that bgzf_copy takes while it flushes a block's bytes (csrc/bgzf_decode.hip), restated in
Python and compared with zlib.crc32: the CRC in its linear form (register starts at 0, the standard's all-ones start = the block's
first four bytes inverted, zero bytes in front of a message change nothing), per-lane COLUMNS of the 2 KiB segments
(acc = later_2K(acc) ^ crc(column's 32 bytes), slicing by 4), the pairwise join of the 64 columns, the move past the tail, and the
tail cut into 32-byte pieces from the block's END.  What htslib checks per block (the reference reads BAM through pysam / htslib:
indexing.py:19); the kernel itself is pinned on the GPU by tests/test_bam_device.py (a changed byte, a changed CRC field, every
intact file's blocks)."""
import random
import zlib

POLY = 0xEDB88320


def _times(mat, vec):
    r, i = 0, 0
    while vec:
        if vec & 1:
            r ^= mat[i]
        vec >>= 1
        i += 1
    return r


def _operators():
    """"append 2^k zero bits" as 32 x 32 matrices over GF(2): one zero bit, squared up (zlib's crc32_combine construction)."""
    cur = [POLY] + [1 << (i - 1) for i in range(1, 32)]
    mats, bits = {}, 1
    while bits <= 8 * 2048:
        mats[bits] = list(cur)
        cur = [_times(cur, cur[i]) for i in range(32)]
        bits <<= 1
    return mats


MATS = _operators()
T0 = []
for _v in range(256):
    _c = _v
    for _ in range(8):
        _c = (_c >> 1) ^ (POLY if _c & 1 else 0)
    T0.append(_c)
T = [T0]
for _k in range(1, 4):
    T.append([T0[c & 255] ^ (c >> 8) for c in T[-1]])      # T[k][v]: the register after byte v and k zero bytes


def later(c, nbytes):
    k = 0
    while nbytes:
        if nbytes & 1:
            c = _times(MATS[8 << k], c)
        nbytes >>= 1
        k += 1
    return c


def crc16(c, b16):
    for k in range(4):
        x = c ^ int.from_bytes(b16[4 * k:4 * k + 4], "little")
        c = T[3][x & 255] ^ T[2][(x >> 8) & 255] ^ T[1][(x >> 16) & 255] ^ T[0][x >> 24]
    return c


def masked(b16, at, frm, inv):
    out = bytearray(16)
    for i in range(16):
        p = at + i
        v = b16[i] if p >= frm else 0
        if p >= frm and inv <= p < inv + 4:
            v ^= 0xFF
        out[i] = v
    return bytes(out)


def fold32(vals):
    c = list(vals)
    for k in range(6):
        d = 1 << k
        right = [c[lane + d] if lane + d < 64 else c[lane] for lane in range(64)]
        c = [later(c[lane], 32 << k) ^ right[lane] for lane in range(64)]
    return c[0]


def block_crc(data, a0):
    """A block whose first byte lies at position a0 (0 .. 15) of its first 16-byte row, as bgzf_copy sees it."""
    ulen, vend = len(data), a0 + len(data)
    ring = bytes(a0) + data
    if ulen < 128:
        return zlib.crc32(data)                              # (the kernel: byte by byte with T[0])
    acc, flushed = [0] * 64, 0
    while vend - flushed >= 2048:
        for lane in range(64):
            p0, p1 = ring[flushed + 32 * lane:flushed + 32 * lane + 16], ring[flushed + 32 * lane + 16:flushed + 32 * lane + 32]
            if flushed == 0:
                p0, p1 = masked(p0, 32 * lane, a0, a0), masked(p1, 32 * lane + 16, a0, a0)
            acc[lane] = later(acc[lane], 2048) ^ crc16(crc16(0, p0), p1)
        flushed += 2048
    frm = max(flushed, a0)
    tail = [0] * 64
    for lane in range(64):
        ps = vend - 32 * (64 - lane)
        if ps + 32 > frm:
            rd = lambda x: bytes(ring[i] if 0 <= i < len(ring) else 0xAA for i in range(x, x + 16))    # noqa: E731 (bytes in front are masked)
            tail[lane] = crc16(crc16(0, masked(rd(ps), ps, frm, a0)), masked(rd(ps + 16), ps + 16, frm, a0))
    full = later(fold32(acc), vend - flushed)
    return (~(full ^ fold32(tail))) & 0xFFFFFFFF


def test_columns_join_and_tail_give_zlibs_crc():
    rng = random.Random(5)
    cases = [(0, 128), (15, 129), (13, 2048 - 13), (3, 2048 - 3 + 1), (0, 4096), (7, 4096 - 7 - 1), (12, 6000), (1, 65280)]
    cases += [(rng.randrange(16), rng.randrange(128, 20000)) for _ in range(12)]
    for a0, ulen in cases:
        data = bytes(rng.getrandbits(8) for _ in range(ulen))
        assert block_crc(data, a0) == zlib.crc32(data), (a0, ulen)
