"""BGZF / BAM writer (SAM spec §4.1, §4.2) for synthetic inputs.  Takes the flat read arrays of
the tcmi_reads layout, so what is written is exactly what the native reader must give back."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


# test hooks: zlib strategy / memLevel of the deflate streams, and a Z_FULL_FLUSH every so many bytes (several deflate blocks,
# dynamic and empty stored ones, inside one BGZF block)
DEFLATE = {"strategy": zlib.Z_DEFAULT_STRATEGY, "mem_level": 8, "flush_every": 0}


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, DEFLATE["mem_level"], DEFLATE["strategy"])
    step = DEFLATE["flush_every"]
    if step:
        body = b"".join(co.compress(data[o:o + step]) + co.flush(zlib.Z_FULL_FLUSH) for o in range(0, len(data), step)) + co.flush()
    else:
        body = co.compress(data) + co.flush()
    bsize = len(body) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bam(path, reads, ref_name="ref", ref_len=0, level=1, sam_text=None, block=0xFF00, split_records=False, refs=None):
    """reads: dict with n_reads,pos,flag,l_qseq,cigar_off,cigar,seq_off,seq[,qual,qual_off,tid,mapq].
    Blocks are cut the way htslib cuts them (bgzf_flush_try): the header gets blocks of its own and a record that does not
    fit into the current block starts the next one, so every BGZF block begins on a record boundary (only a record larger
    than a block spans several).  split_records=True fills every block to the brim instead (records straddle blocks, as
    some other writers do).  refs: [(name, length), ...] for more than one @SQ."""
    n = int(reads["n_reads"])
    pos, flag, lq = reads["pos"], reads["flag"], reads["l_qseq"]
    co, cg, so, sq = reads["cigar_off"], reads["cigar"], reads["seq_off"], reads["seq"]
    qual = reads.get("qual")
    qoff = reads.get("qual_off")
    if qual is not None and qoff is None:
        qoff = np.concatenate(([0], np.cumsum(np.asarray(lq, np.int64))))
    tid = reads.get("tid")
    refs = refs or [(ref_name, ref_len)]
    text = sam_text if sam_text is not None else "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    tb = text.encode()
    out = bytearray(b"BAM\1" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(refs)))
    for name, ln in refs:
        nb = name.encode() + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
    with open(path, "wb") as fh:
        def flush(final=False):
            nonlocal out
            while len(out) >= block or (final and out):
                fh.write(_bgzf_block(bytes(out[:block]), level))
                out = out[block:]
        if not split_records:
            flush(final=True)                       # the header never shares a block with records
        cg_b = np.ascontiguousarray(cg, "<u4").tobytes()
        sq_b = np.ascontiguousarray(sq, np.uint8).tobytes()
        q_b = np.ascontiguousarray(qual, np.uint8).tobytes() if qual is not None else None
        nm_off, nm = reads.get("name_off"), reads.get("names")
        nm_b = bytes(bytearray(nm)) if nm is not None else None
        mt, mp, tl = reads.get("next_tid"), reads.get("next_pos"), reads.get("tlen")
        for i in range(n):
            c0, c1 = int(co[i]), int(co[i + 1])
            s0 = int(so[i])
            l = int(lq[i])
            nc = c1 - c0
            span = 0
            for k in range(c0, c1):
                if (int(cg[k]) & 0xF) in (0, 2, 3, 7, 8):
                    span += int(cg[k]) >> 4
            name = (nm_b[int(nm_off[i]):int(nm_off[i + 1])] + b"\0") if nm_b is not None and nm_off is not None else b"r%d\0" % i
            t = int(tid[i]) if tid is not None else 0
            p = int(pos[i])
            q = q_b[int(qoff[i]):int(qoff[i]) + l] if q_b is not None else b"\xff" * l
            rec = (struct.pack("<iiBBHHHIiii", t, p, len(name), 60, _reg2bin(max(p, 0), max(p, 0) + max(span, 1)), nc,
                               int(flag[i]), l, int(mt[i]) if mt is not None else -1, int(mp[i]) if mp is not None else -1,
                               int(tl[i]) if tl is not None else 0) + name + cg_b[4 * c0:4 * c1] +
                   sq_b[s0:s0 + (l + 1) // 2] + q)
            if split_records:
                out += struct.pack("<i", len(rec)) + rec
                if len(out) >= 4 * block:
                    flush()
            else:
                if out and len(out) + 4 + len(rec) > block:
                    flush(final=True)
                out += struct.pack("<i", len(rec)) + rec
                if len(out) >= block:               # a record larger than a block: spans several, the next record starts afresh
                    flush(final=True)
        flush(final=True)
        fh.write(_EOF)


def write_bam_fast(path, pos, flag, seq_packed, read_len, ref_name="ref", ref_len=0, level=1, qual=30, part=(True, True), first_id=0, names=None):
    """Vectorised writer for the bench workload: n reads, all `read_len`M, flags from `flag`,
    seq_packed uint8 [n, ceil(read_len/2)] in BAM nibble order, constant quality.  part = (first, last): a large file is written
    in several calls, slice by slice (the header goes with the first, the end-of-file block with the last); first_id numbers the names.
    qual: one value, or uint8 [n, read_len]; names: uint8 [n, k] (k characters each, no NUL) instead of the seven-digit numbers."""
    n = len(pos)
    nb = (read_len + 1) // 2
    name_len = 8 if names is None else names.shape[1] + 1     # fixed-width names: 7 digits (or the caller's characters) + NUL
    rec_len = 32 + name_len + 4 + nb + read_len
    rec = np.zeros((n, 4 + rec_len), np.uint8)
    def put(col, arr, dt):
        a = np.ascontiguousarray(arr, dt).view(np.uint8).reshape(n, -1)
        rec[:, col:col + a.shape[1]] = a
    put(0, np.full(n, rec_len), "<i4")
    put(4, np.zeros(n), "<i4")                      # refID
    put(8, pos, "<i4")
    rec[:, 12] = name_len
    rec[:, 13] = 60
    beg = np.asarray(pos, np.int64)
    end = beg + read_len - 1
    binv = np.zeros(n, np.int64)
    done = np.zeros(n, bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (end >> shift))
        binv[hit] = base + (beg[hit] >> shift)
        done |= hit
    put(14, binv, "<u2")
    put(16, np.full(n, 1), "<u2")                   # n_cigar_op
    put(18, flag, "<u2")
    put(20, np.full(n, read_len), "<u4")
    put(24, np.full(n, -1), "<i4")
    put(28, np.full(n, -1), "<i4")
    put(32, np.zeros(n), "<i4")
    if names is None:
        ids = (np.arange(n) + first_id) % 10000000
        digits = np.zeros((n, 7), np.uint8)
        for k in range(7):
            digits[:, 6 - k] = 48 + (ids // 10 ** k) % 10
        rec[:, 36:43] = digits
    else:
        rec[:, 36:36 + names.shape[1]] = names
    put(36 + name_len, np.full(n, (read_len << 4) | 0), "<u4")
    s0 = 36 + name_len + 4
    rec[:, s0:s0 + nb] = seq_packed
    rec[:, s0 + nb:s0 + nb + read_len] = qual
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (ref_name, ref_len)
    tb, nmb = text.encode(), ref_name.encode() + b"\0"
    head = (b"BAM\1" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", 1) + struct.pack("<i", len(nmb)) + nmb +
            struct.pack("<i", ref_len))
    block = 0xFF00
    per = max(1, block // (4 + rec_len))            # whole records per block, as htslib cuts them
    with open(path, "wb" if part[0] else "ab") as fh:
        if part[0]:
            for o in range(0, len(head), block):
                fh.write(_bgzf_block(head[o:o + block], level))
        flat = rec.reshape(-1)
        step = per * (4 + rec_len)
        for o in range(0, flat.size, step):
            fh.write(_bgzf_block(flat[o:o + step].tobytes(), level))
        if part[1]:
            fh.write(_EOF)
