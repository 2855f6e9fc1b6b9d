"""A BAM file assembled by hand in this test from the SAM specification (§4.1 BGZF, §4.2 BAM), byte by byte with
struct.pack — not by trueconsense_amd.io.bamwriter — with everything the synthetic writers never produce: two @SQ lines,
aux tags of several types behind the qualities, read names of different lengths, a multi-op CIGAR, an unmapped tail,
BGZF blocks with an extra gzip subfield in front of BC, a stored (uncompressed) deflate block, an empty block in the
middle, and (second file) records that straddle block boundaries.  The host reader, the oracle's reader and (GPU) the
device decoder must all give the same reads."""
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import c_oracle
from trueconsense_amd import engine

NT16 = "=ACMGRSVTWYHKDBN"
OPS = "MIDNSHP=X"


def rec(tid, pos, name, flag, cigar, seq, qual, aux=b"", mapq=37, mtid=-1, mpos=-1, tlen=0):
    cg = b"".join(struct.pack("<I", (n << 4) | OPS.index(op)) for n, op in cigar)
    l = len(seq)
    codes = [NT16.index(c) for c in seq] + ([0] if l & 1 else [])
    sq = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    nm = name.encode() + b"\0"
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(nm), mapq, 4680, len(cigar), flag, l, mtid, mpos, tlen) + nm + cg + sq + bytes(qual) + aux
    return struct.pack("<i", len(body)) + body


def bgzf(data, level=6, extra_subfield=False, stored=False):
    if stored:
        body = b"\x01" + struct.pack("<HH", len(data), len(data) ^ 0xFFFF) + data      # BFINAL=1, BTYPE=00
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(data) + co.flush()
    xtra = (b"XY" + struct.pack("<H", 3) + b"abc" if extra_subfield else b"")            # another subfield in front of BC
    xlen = len(xtra) + 6
    bsize = 12 + xlen + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", xlen) + xtra + b"BC\x02\x00" + struct.pack("<H", bsize - 1) +
            body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def build(path, straddle):
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrA\tLN:500\n@SQ\tSN:chrB\tLN:300\n@PG\tID:hand\n".encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for name, ln in (("chrA", 500), ("chrB", 300)):
        nb = name.encode() + b"\0"
        head += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
    aux = b"NMC\x02" + b"MDZ10A5\0" + b"ASs" + struct.pack("<h", -7) + b"XBBi" + struct.pack("<ii", 2, 11) + struct.pack("<i", -3)
    reads = [
        rec(0, 3, "r1", 0, [(20, "M")], "ACGTACGTACGTACGTACGT", [30] * 20, aux),
        rec(0, 3, "read_with_a_longer_name/1", 99, [(2, "S"), (8, "M"), (2, "I"), (6, "M"), (3, "D"), (5, "M"), (1, "H")],
            "NNACGTACGTGGACGTACCCCCC"[:23], list(range(10, 33)), b"", mtid=0, mpos=40, tlen=80),
        rec(0, 10, "r3", 16, [(5, "="), (1, "X"), (9, "M")], "ACGTAGACGTNRACG", [2] * 15, b"RGZgrp\0"),
        rec(0, 40, "read_with_a_longer_name/1", 147, [(15, "M")], "TTTTTGGGGGCCCCC", [40] * 15, b"", mtid=0, mpos=3, tlen=-80),
        rec(0, 100, "skip", 0, [(4, "M"), (50, "N"), (4, "M")], "ACGTACGT", [25] * 8),
        rec(0, 480, "past_the_end", 0, [(30, "M")], "ACGTAC" * 5, [20] * 30),
        rec(0, 120, "placed_unmapped", 4 | 1 | 64, [], "ACGT", [9] * 4, mtid=0, mpos=120),
        rec(-1, -1, "u1", 4, [], "ACGTN", [1, 2, 3, 4, 5]),
        rec(-1, -1, "u2_no_seq", 4, [], "", []),
    ]
    # sort order of a coordinate-sorted BAM: (tid, pos), unplaced last
    order = [0, 1, 2, 3, 4, 6, 5, 7, 8]
    reads = [reads[i] for i in order]
    with open(path, "wb") as fh:
        fh.write(bgzf(head, extra_subfield=True))
        if not straddle:
            fh.write(bgzf(b"".join(reads[:2]), level=9))
            fh.write(bgzf(b""))                                     # an empty block between two data blocks
            fh.write(bgzf(b"".join(reads[2:5]), stored=True))       # deflate "stored" block
            fh.write(bgzf(b"".join(reads[5:]), level=1, extra_subfield=True))
        else:
            blob = b"".join(reads)
            cuts = [0, 37, 38, 120, 121, 300, len(blob)]            # through block_size fields and record bodies
            for a, b in zip(cuts[:-1], cuts[1:]):
                fh.write(bgzf(blob[a:b], level=6))
        fh.write(EOF_BLOCK)
    return order


EXPECT = {
    "pos": [3, 3, 10, 40, 100, 120, 480, -1, -1], "tid": [0, 0, 0, 0, 0, 0, 0, -1, -1],
    "flag": [0, 99, 16, 147, 0, 4 | 1 | 64, 0, 4, 4], "l_qseq": [20, 23, 15, 15, 8, 4, 30, 5, 0],
    "n_cigar": [1, 7, 3, 1, 3, 0, 1, 0, 0],
}


def check_arrays(a):
    for k in ("pos", "tid", "flag", "l_qseq"):
        assert list(map(int, a[k])) == EXPECT[k], k
    assert np.diff(a["cigar_off"].astype(np.int64)).tolist() == EXPECT["n_cigar"]
    assert [int(x) for x in a["cigar"][1:8]] == [(2 << 4) | 4, (8 << 4) | 0, (2 << 4) | 1, (6 << 4) | 0, (3 << 4) | 2, (5 << 4) | 0, (1 << 4) | 5]
    s0 = int(a["seq_off"][2])
    assert [int(x) for x in a["seq"][s0:s0 + 8]] == [0x12, 0x48, 0x14, 0x12, 0x48, 0xF5, 0x12, 0x40]     # ACGTAGACGTNRACG
    q0 = int(a["qual_off"][1]) if "qual_off" in a else 20
    assert [int(x) for x in a["qual"][q0:q0 + 23]] == list(range(10, 33))


@pytest.mark.parametrize("straddle", [False, True])
def test_hand_assembled_bam_host_reader_and_oracle(tmp_path, straddle):
    p = str(tmp_path / "hand.bam")
    build(p, straddle)
    o = c_oracle.read_bam(p)
    assert (o["n_ref"], o["ref0_name"], o["ref0_len"], o["n_reads"]) == (2, "chrA", 500, 9)
    check_arrays(o)
    b = engine.BamFile(p, threads=3)
    assert (b.nreferences, b.references, b.lengths, b.n_reads, b.sorted) == (2, ("chrA",), (500,), 9, 1)
    assert "@PG\tID:hand" in b.text
    a = b.arrays()
    check_arrays(a)
    for k in ("pos", "flag", "l_qseq", "tid", "cigar_off", "seq_off"):
        assert np.array_equal(np.asarray(a[k])[:len(o[k])], o[k]), k
    assert np.array_equal(a["cigar"][:len(o["cigar"])], o["cigar"]) and np.array_equal(a["seq"][:len(o["seq"])], o["seq"])
    assert np.array_equal(a["qual"][:len(o["qual"])], o["qual"])
    assert engine.reads_extent(b, 500) == 510                          # the read at 480 runs 10 positions past chrA's end


@pytest.mark.gpu
def test_hand_assembled_bam_on_the_device(tmp_path):
    from trueconsense_amd import _ffi, _state
    ctx = _state.default_context()
    p = str(tmp_path / "hand.bam")
    build(p, straddle=False)
    o = c_oracle.read_bam(p)
    want = c_oracle.tally(o, 510)
    d = engine.DeviceBam(p)
    assert (d.nreferences, d.references, d.lengths) == (2, ("chrA",), (500,))
    stream, rec_off = d.decode_to_host(ctx)
    assert len(rec_off) == 9 and bytes(stream[:4]) == b"BAM\x01"
    rs = ctx.upload_bamfile(d)
    assert rs.n_reads == 9 and rs.n_piled == 6 and rs.max_end == 510
    got = ctx.step(rs, 510, 1, True)[3]
    assert np.array_equal(got, want)
    assert got[3 + 2 + 8 - 1 - 2, 6] == 1 and got[:, 5].sum() == 3 and got[104:154, 0].sum() == 50     # I after the 8th matched base; 3 D; the N skip covers
    rs.free()
    d.close()
    # ... and the host reader's arrays through the device packer give the same matrix
    b = engine.BamFile(p)
    assert np.array_equal(ctx.tally(b, L=510), want)
    # the straddling variant (blocks cut through block_size fields and record bodies): on the device as well
    p2 = str(tmp_path / "hand2.bam")
    build(p2, straddle=True)
    d2 = engine.DeviceBam(p2)
    stream2, rec2 = d2.decode_to_host(ctx)
    assert np.array_equal(stream2, stream) and np.array_equal(rec2, rec_off)
    rs2 = ctx.upload_bamfile(d2)
    assert rs2.n_reads == 9 and np.array_equal(ctx.step(rs2, 510, 1, True)[3], want)
    rs2.free()
    d2.close()
    assert np.array_equal(ctx.tally(engine.BamFile(p2), L=510), want)


def test_cigar_in_cg_tag_is_refused(tmp_path):
    """SAM spec §4.2.2: more than 65 535 CIGAR operations live in a CG:B,I tag behind the placeholder <l_seq>S<n>N; htslib
    swaps the real CIGAR in.  The same placeholder WITHOUT the tag is an ordinary (if odd) read."""
    from trueconsense_amd import _ffi
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:100\n"
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\0" + struct.pack("<i", 100)
    cg_tag = b"CGBI" + struct.pack("<i", 2) + struct.pack("<II", (4 << 4) | 0, (4 << 4) | 0)
    for aux, ok in ((b"", True), (b"NMC\x01" + cg_tag, False)):
        p = str(tmp_path / ("cg%d.bam" % ok))
        with open(p, "wb") as fh:
            fh.write(bgzf(head))
            fh.write(bgzf(rec(0, 5, "x", 0, [(8, "S"), (8, "N")], "ACGTACGT", [30] * 8, aux)))
            fh.write(EOF_BLOCK)
        if ok:
            assert engine.BamFile(p).n_reads == 1
        else:
            with pytest.raises(_ffi.TcmiError) as e:
                engine.BamFile(p)
            assert e.value.code == _ffi.E_UNSUPPORTED and "CG tag" in str(e.value)


def forged(path, what):
    """A file whose CRC-32 and record chain are in order but whose second record lies about its variable-length fields."""
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:400\n"
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\0" + struct.pack("<i", 400)
    good = rec(0, 5, "ok", 0, [(20, "M")], "ACGTACGTACGTACGTACGT", [30] * 20)
    r = bytearray(rec(0, 9, "liar", 0, [(4, "S"), (16, "M")], "ACGTACGTACGTACGTACGT", [30] * 20))
    if what == "l_seq":
        struct.pack_into("<I", r, 4 + 16, 1 << 30)                  # bases that are not there (a large S or I op would walk off the stream)
        struct.pack_into("<I", r, 4 + 32 + 5, ((1 << 28) - 1) << 4 | 1)      # 268435455I
    elif what == "n_cigar":
        struct.pack_into("<H", r, 4 + 12, 60000)                    # CIGAR operations that are really SEQ, QUAL and the next records
    else:
        r[4 + 8] = 0                                                # l_read_name = 0
    tail = [rec(0, 30 + k, "t%d" % k, 0, [(20, "M")], "ACGTACGTACGTACGTACGT", [30] * 20) for k in range(40)]
    with open(path, "wb") as fh:
        fh.write(bgzf(head))
        fh.write(bgzf(good + bytes(r) + b"".join(tail)))
        fh.write(EOF_BLOCK)


@pytest.mark.parametrize("what", ["l_seq", "n_cigar", "l_read_name"])
def test_record_fields_that_overrun_block_size_host(tmp_path, what):
    from trueconsense_amd import _ffi
    p = str(tmp_path / "forged.bam")
    forged(p, what)
    with pytest.raises(_ffi.TcmiError) as e:
        engine.BamFile(p)
    assert e.value.code == _ffi.E_FORMAT


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["l_seq", "n_cigar", "l_read_name"])
def test_record_fields_that_overrun_block_size_device(tmp_path, what):
    """The device decoder checks every record's l_read_name / n_cigar_op / l_seq against its block_size before any kernel follows
    them (the CRC-32 cannot tell: the file is intact, it just is not a BAM file): refused, with the CRC check on and off."""
    from trueconsense_amd import _ffi, _state
    ctx = _state.default_context()
    p = str(tmp_path / "forged.bam")
    forged(p, what)
    try:
        for crc in (1, 0):
            ctx.set_option("verify_crc", crc)
            d = engine.DeviceBam(p)
            with pytest.raises(_ffi.TcmiError) as e:
                ctx.upload_bamfile(d)
            assert e.value.code in (_ffi.E_UNSUPPORTED, _ffi.E_FORMAT)
            d.close()
    finally:
        ctx.set_option("verify_crc", 1)
