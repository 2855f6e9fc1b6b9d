"""GPU: the one-sync file path (pk_index + pk_place + pk_pack: record index, record chain, classification and the places of
records and kept reads from per-block aggregates every workgroup adds up for itself, no host round trip; tcmi_bamfile_step:
decode, pack, tally and call queued back to back) against the several-kernel path (three + one waits) and against the oracle.  Integer / byte work: bit-exact.  What the reference
computes here is indexing.BuildIndex (indexing.py:75-154) and the position-local part of BuildConsensus (Sequences.py:119-165)."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from tests import fuzz_reads as fz
from trueconsense_amd import _ffi, engine
from trueconsense_amd import synthetic as sy
from trueconsense_amd.io import bamwriter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def both_paths(ctx, path, L, expect_fused=True, blocks=None):
    """The file (or a range of its blocks) through both paths -> the read sets' figures and the step's outputs must be identical."""
    d = engine.DeviceBam(path)
    out = []
    for one_sync in (1, 0):
        ctx.set_option("one_sync", one_sync)
        t0, d0 = ctx.stat("one_sync_taken"), ctx.stat("one_sync_declined")
        rs = ctx.upload_bamfile(d, blocks=blocks)
        if one_sync:
            took = ctx.stat("one_sync_taken") - t0
            assert took == (1 if expect_fused else 0), "one-sync path: taken %d, declined %d (flags 0x%x)" % (
                took, ctx.stat("one_sync_declined") - d0, ctx.stat("one_sync_last_decline_flags"))
        else:
            assert ctx.stat("one_sync_taken") == t0
        Lx = max(L, rs.max_end, 1)
        plain, alt, flags, counts = ctx.step(rs, Lx, 30, True)
        out.append((rs.n_reads, rs.n_piled, rs.algorithmic_bytes, rs.max_end, Lx, plain, alt, flags, counts))
        rs.free()
    ctx.set_option("one_sync", 1)
    d.close()
    a, b = out
    assert a[:5] == b[:5], (a[:5], b[:5])
    for x, y in zip(a[5:], b[5:]):
        assert np.array_equal(x, y), np.argwhere(x != y)[:5]
    return a


def write(tmp_path, name, reads, ref_name, L, **kw):
    p = str(tmp_path / name)
    bamwriter.write_bam(p, reads, ref_name, L, **kw)
    return p


def test_one_sync_path_equals_the_several_kernel_path_and_the_oracle(ctx, tmp_path):
    ref, orfs = sy.make_reference()
    L = len(ref)
    rng = np.random.default_rng(21)
    cases = []
    n = 120_000
    r = sy.make_reads(ref, n, seed=31)
    p = str(tmp_path / "plain.bam")
    bamwriter.write_bam_fast(p, r["pos"], r["flag"], r["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)
    cases.append((p, L))
    cases.append((write(tmp_path, "indel.bam", sy.make_reads(ref, 60_000, seed=32, indel_sites=sy.default_indel_sites(orfs)), "MN908947.3", L), L))
    cases.append((write(tmp_path, "fuzz.bam", fz.random_reads(rng, 8000, 3000), "ref", 3000, level=4), 3000))       # every CIGAR op, N bases, unmapped reads
    # short reads: ~1 200 records per BGZF block (several tiles of 256 per workgroup), and tiny blocks (a block per few records)
    cases.append((write(tmp_path, "short.bam", sy.make_reads(ref, 90_000, read_len=30, seed=33), "MN908947.3", L), L))
    for path, Lc in cases:
        got = both_paths(ctx, path, Lc)
        reads = c_oracle.read_bam(path)
        want = c_oracle.tally(reads, got[4])
        assert got[0] == reads["n_reads"]
        assert np.array_equal(got[8], want), np.argwhere(got[8] != want)[:5]
        wp, wa, wf = c_oracle.call(want, 30, True)
        assert np.array_equal(got[5], wp) and np.array_equal(got[6], wa) and np.array_equal(got[7], wf)


def test_one_sync_path_on_block_ranges_and_straddling_records(ctx, tmp_path):
    """Block ranges (the record chain starts OPEN: wherever the range's first block finds a record) add up to the file; records
    that straddle BGZF blocks (every block filled to the brim; tiny blocks: a record over several) go through the blocks'
    chain functions."""
    ref, orfs = sy.make_reference(L=6000, cds=[(100, 2500), (3000, 5800)])
    L = len(ref)
    reads = sy.make_reads(ref, 30_000, seed=41, indel_sites=sy.default_indel_sites(orfs))
    want = c_oracle.tally(reads, L)
    for name, kw in (("brim.bam", dict(split_records=True)), ("tiny.bam", dict(split_records=True, block=1021)), ("htslib.bam", {})):
        p = write(tmp_path, name, reads, "ref", L, **kw)
        whole = both_paths(ctx, p, L)
        assert np.array_equal(whole[8], want)
        d = engine.DeviceBam(p)
        nb = d.n_blocks
        d.close()
        acc = np.zeros_like(want)
        cuts = sorted(set([0, nb // 3, nb // 2, nb - 1, nb]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = both_paths(ctx, p, L, blocks=(a, b - a), expect_fused=b > a)
            acc += part[8]
        assert np.array_equal(acc, want)


def test_compressed_bytes_in_pieces_ahead_of_the_decoder(ctx, tmp_path):
    """Option "h2d_pieces": the compressed bytes cross PCIe in pieces of whole blocks on a copy stream, each piece's blocks inflated as
    soon as it has arrived (a rank's range of a large file does so by itself from 12 MB on).  Forced here on small files — blocks cut
    on record boundaries, filled to the brim, tiny; the whole file and ranges; 2 .. 16 pieces: the same counts as in one copy."""
    ref, orfs = sy.make_reference(L=6000, cds=[(100, 2500), (3000, 5800)])
    L = len(ref)
    reads = sy.make_reads(ref, 40_000, seed=43, indel_sites=sy.default_indel_sites(orfs))
    want = c_oracle.tally(reads, L)
    try:
        for name, kw in (("hts.bam", {}), ("brim.bam", dict(split_records=True)), ("tiny.bam", dict(split_records=True, block=1021))):
            p = write(tmp_path, name, reads, "ref", L, **kw)
            d = engine.DeviceBam(p)
            nb = d.n_blocks
            d.close()
            for pieces in (2, 3, 7, 16):
                ctx.set_option("h2d_pieces", pieces)
                before = ctx.stat("h2d_piped")
                whole = both_paths(ctx, p, L)
                assert np.array_equal(whole[8], want), (name, pieces)
                assert ctx.stat("h2d_piped") - before == 2 if nb >= 64 else True        # (both paths decode: twice)
                acc = np.zeros_like(want)
                for a, b in ((0, nb // 2), (nb // 2, nb - nb // 2)):
                    acc += both_paths(ctx, p, L, blocks=(a, b))[8]
                assert np.array_equal(acc, want), (name, pieces)
        assert ctx.stat("h2d_piped") > 0
    finally:
        ctx.set_option("h2d_pieces", 0)


def test_bamfile_step_is_one_call_and_exact(ctx, tmp_path):
    ref, orfs = sy.make_reference()
    L = len(ref)
    reads = sy.make_reads(ref, 200_000, seed=51, indel_sites=sy.default_indel_sites(orfs))
    p = write(tmp_path, "step.bam", reads, "MN908947.3", L)
    want = c_oracle.tally(reads, L)
    wp, wa, wf = c_oracle.call(want, 30, True)
    d = engine.DeviceBam(p)
    for resident in (False, True):
        if resident:
            d.to_device(ctx)
        t0 = ctx.stat("one_sync_taken")
        rs, plain, alt, flags, counts = ctx.bamfile_step(d, L, 30, True)
        assert ctx.stat("one_sync_taken") == t0 + 1, "flags 0x%x" % ctx.stat("one_sync_last_decline_flags")
        assert rs.n_reads == reads["n_reads"] and rs.max_end <= L
        assert np.array_equal(counts, want), np.argwhere(counts != want)[:5]
        assert np.array_equal(plain, wp) and np.array_equal(alt, wa) and np.array_equal(flags, wf)
        # the stream stays resident: insert tokens of the candidate columns, as after the several-kernel path
        cand = [int(i) + 1 for i in np.flatnonzero(flags & _ffi.F_INSCAND)]
        if cand:
            got = ctx.readset_modal_tokens(rs, cand)
            host = engine.modal_tokens(reads, cand)
            assert got == host
        # without the counts (the runner's usual call), twice in a row on the same context (the call kernel leaves the matrix zeroed)
        rs.free()
        for _ in range(2):
            rs2, p2, a2, f2, c2 = ctx.bamfile_step(d, L, 30, True, want_counts=False)
            assert c2 is None and np.array_equal(p2, wp) and np.array_equal(a2, wa) and np.array_equal(f2, wf)
            rs2.free()
    d.close()


def test_bamfile_step_when_reads_reach_beyond_ref_len_or_are_long(ctx, tmp_path):
    """ref_len smaller than the reads' extent: the step is queued for ref_len positions and run again for the extent; reads of
    more than 512 positions are left to tally_stream_kernel, which the second step launches."""
    ref, _ = sy.make_reference(L=5000, cds=[(100, 2000)])
    reads = sy.make_reads(ref, 20_000, seed=61)
    p = write(tmp_path, "over.bam", reads, "ref", 5000)
    want = c_oracle.tally(reads, 5000)
    d = engine.DeviceBam(p)
    rs, plain, alt, flags, counts = ctx.bamfile_step(d, 3000, 30, True)
    assert len(plain) == rs.max_end > 3000
    assert np.array_equal(counts, want[:rs.max_end])
    rs.free()
    d.close()
    long_reads = fz.random_reads(np.random.default_rng(62), 3000, 5000, long_reads=True)
    p = write(tmp_path, "long.bam", long_reads, "ref", 5000)
    d = engine.DeviceBam(p)
    rs, plain, alt, flags, counts = ctx.bamfile_step(d, 5000, 30, True)
    Lx = max(5000, rs.max_end)
    assert len(plain) == Lx
    assert np.array_equal(counts, c_oracle.tally(long_reads, Lx))
    rs.free()
    d.close()


def test_damaged_files_are_refused_as_before(ctx, tmp_path):
    """A file the one-sync path cannot vouch for goes to the several-kernel path, which words the refusal: same errors."""
    ref, _ = sy.make_reference(L=4000, cds=[(100, 2000)])
    reads = sy.make_reads(ref, 20_000, seed=71)
    p = write(tmp_path, "ok.bam", reads, "ref", 4000)
    raw = bytearray(open(p, "rb").read())
    bad = str(tmp_path / "bad.bam")
    # one changed byte in the middle of a deflate payload of a block in the middle of the file
    raw[len(raw) // 2] ^= 0x5A
    open(bad, "wb").write(bytes(raw))
    msgs = []
    for one_sync in (1, 0):
        ctx.set_option("one_sync", one_sync)
        try:
            d = engine.DeviceBam(bad)
        except _ffi.TcmiError as e:                                 # (the damage hit a block header: the host's block walk refuses the file)
            msgs.append(("read", e.code))
            continue
        try:
            rs = ctx.upload_bamfile(d)
            got = ctx.step(rs, 4000, 30, True)[3]
            msgs.append(("ok", got.tobytes()))
            rs.free()
        except _ffi.TcmiError as e:
            msgs.append((e.code, str(e)))
        d.close()
    ctx.set_option("one_sync", 1)
    assert msgs[0] == msgs[1]
    assert ctx.stat("one_sync_declined") >= 1 or msgs[0][0] in ("read", "ok")


def _stream_and_records(path):
    """The file's inflated stream and, per alignment record, (start, end, start of its QUAL) — by zlib."""
    import gzip
    import struct
    s = gzip.open(path).read()
    l_text, = struct.unpack_from("<i", s, 4)
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", s, at)
    at += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", s, at)
        at += 4 + l_name + 4
    recs = []
    while at < len(s):
        bs, = struct.unpack_from("<i", s, at)
        l_name = s[at + 12]
        n_cig, = struct.unpack_from("<H", s, at + 16)
        l_seq, = struct.unpack_from("<i", s, at + 20)
        recs.append((at, at + 4 + bs, at + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2))
        at += 4 + bs
    return s, recs


def test_block_ranges_must_join_into_one_chain_of_records(ctx, tmp_path):
    """A block range in the middle of a file starts at the first offset that LOOKS like an alignment record.  Here one read's
    qualities spell a complete record header (every field plausible, its block_size ending where the real record ends) right where
    a BGZF block begins: a range that starts with that block takes it for a read and carries on in step with the true chain —
    nothing inside the range can tell.  The range in front knows better (its last record ends further on): the ranges' anchors
    (tcmi_readset_range_anchors) do not join, distributed.check_range_anchors and tcmi_split_step's anchor word refuse the file."""
    import ctypes as C
    import struct

    import torch

    from trueconsense_amd import distributed as td
    ref, _ = sy.make_reference(L=4000, cds=[(100, 2000)])
    L = len(ref)
    n = 3000
    reads = sy.make_reads(ref, n, seed=81)
    assert int(reads["l_qseq"].min()) == 150 == int(reads["l_qseq"].max())
    reads["qual"] = np.full(n * 150, 30, np.uint8)
    B = 4093
    p0 = write(tmp_path, "plain.bam", reads, "ref", L, split_records=True, block=B)
    s0, recs = _stream_and_records(p0)
    # a block boundary X inside a record's QUAL with 56 bytes of it to spare
    pick = next((i, x) for i, (a, e, q) in enumerate(recs) for x in [(q + B - 1) // B * B] if i > 200 and q <= x and x + 56 <= e)
    i, X = pick
    E = recs[i][1]
    fake = struct.pack("<iiiBBHHHiiii", E - X - 4, 0, 100, 1, 0, 0, 1, 0, 10, -1, -1, 0) + b"\0" + struct.pack("<I", 10 << 4) + b"\x11" * 5 + b"\x1e" * 10
    assert len(fake) == 56
    at = i * 150 + (X - recs[i][2])
    reads["qual"][at:at + 56] = np.frombuffer(fake, np.uint8)
    p = write(tmp_path, "trap.bam", reads, "ref", L, split_records=True, block=B)
    s1, recs1 = _stream_and_records(p)
    assert recs1 == recs and s1[X:X + 56] == fake and len(s1) == len(s0)
    want = c_oracle.tally(reads, L)
    d0, d = engine.DeviceBam(p0), engine.DeviceBam(p)
    nb, k = d.n_blocks, X // B
    assert nb > k + 2 and d0.n_blocks == nb

    def ranges(dbam, cut):
        out, acc = [], np.zeros_like(want)
        for a, c in ((0, cut), (cut, nb - cut)):
            rs = ctx.upload_bamfile(dbam, blocks=(a, c))
            out.append((a, c) + rs.range_anchors)
            acc += ctx.step(rs, L, 30, True)[3]
            rs.free()
        return out, acc

    for one_sync in (1, 0):
        ctx.set_option("one_sync", one_sync)
        # wherever block k has a block in front of it in the same decode, the chain does not close there: the file as a whole and a
        # range that holds block k but does not start with it are refused (the host reader's)
        for blocks in (None, (0, k + 1), (k - 1, 3)):
            with pytest.raises(_ffi.TcmiError) as e:
                ctx.upload_bamfile(d, blocks=blocks)
            assert e.value.code == _ffi.E_UNSUPPORTED and "does not close at BGZF block" in str(e.value)
        # ... but a range that STARTS with block k cannot know: both ranges decode, and only their anchors tell
        r, acc = ranges(d, k)
        assert r[0][3] == E and r[1][2] == X, r
        assert not np.array_equal(acc, want)                         # (what the fake read would have cost: counts that are not the file's)
        why = td.check_range_anchors(r, d.inflated_bytes)
        assert why and "starts a record at stream offset %d" % X in why
        for cut in (k - 1, k, k + 1, 1, nb - 1):                     # the same file without the trap: every cut joins
            r, acc = ranges(d0, cut)
            assert td.check_range_anchors(r, d0.inflated_bytes) is None, (cut, r)
            assert np.array_equal(acc, want)
    ctx.set_option("one_sync", 1)

    # tcmi_split_step: the ranges' table rides in the reduce (two "ranks" one after the other on this GPU; the hook adds the first one's share)
    ld = (L + 255) // 256 * 256
    lib = _ffi.lib()

    def split(dbam, cut):
        share = {}
        bufs = [torch.zeros(7 * ld + 13, dtype=torch.int32, device="cuda") for _ in range(2)]   # TCMI_SPLIT_TAIL_WORDS(2)

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
        def keep(user, ptr, nw, stream):
            assert nw == 7 * ld + 13
            torch.cuda.synchronize()
            share["t"] = bufs[1].clone()
            return 0

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
        def add(user, ptr, nw, stream):
            torch.cuda.synchronize()
            bufs[0] += share["t"]
            torch.cuda.synchronize()
            return 0
        out = []
        for rank, (a, c, hook) in ((1, (cut, nb - cut, keep)), (0, (0, cut, add))):
            h, p_, a_, f_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            rc = lib.tcmi_split_step(ctx.handle, dbam.handle, a, c, L, ld, C.c_void_p(bufs[rank].data_ptr()), 30, 1, hook, None, rank, 2,
                                     C.byref(h), C.byref(p_), C.byref(a_), C.byref(f_))
            msg = (lib.tcmi_last_error(ctx.handle) or b"").decode()
            fl = None
            if rc == 0:
                if rank == 0:
                    fl = np.empty(L, np.uint8)
                    C.memmove(fl.ctypes.data, f_, L)
                lib.tcmi_readset_free(ctx.handle, h)
            out.append((rc, msg, fl))
        return out

    wf = c_oracle.call(want, 30, True)[2]
    (rc1, _, _), (rc0, msg0, fl0) = split(d0, k)
    assert rc1 == 0 and rc0 == 0 and np.array_equal(fl0, wf), (rc1, rc0, msg0)
    (rc1, _, _), (rc0, msg0, fl0) = split(d, k)
    assert rc1 == 0 and rc0 == _ffi.E_UNSUPPORTED and "do not join" in msg0, (rc0, msg0)
    d.close()
    d0.close()


def test_prefix_kernels_for_files_of_very_many_blocks(ctx, tmp_path):
    """From 16 384 blocks on the sums in front of a block (records, kept reads, plane words) come from three scan launches instead of
    every workgroup adding them up for itself ("prefix_kernels"): forced on here for small files — whole, as ranges, with records
    across blocks — against the several-kernel path and the oracle."""
    ref, orfs = sy.make_reference(L=6000, cds=[(100, 2500), (3000, 5800)])
    L = len(ref)
    reads = sy.make_reads(ref, 40_000, seed=91, indel_sites=sy.default_indel_sites(orfs))
    want = c_oracle.tally(reads, L)
    try:
        ctx.set_option("prefix_kernels", 1)
        for name, kw in (("brim.bam", dict(split_records=True, block=3001)), ("htslib.bam", dict(block=2500))):       # ~ 3 000 - 4 000 blocks: several tiles of the scan
            p = write(tmp_path, name, reads, "ref", L, **kw)
            whole = both_paths(ctx, p, L)
            assert np.array_equal(whole[8], want)
            d = engine.DeviceBam(p)
            nb = d.n_blocks
            d.close()
            assert nb > 2048
            acc = np.zeros_like(want)
            for a, b in ((0, nb // 3), (nb // 3, nb - nb // 3)):
                acc += both_paths(ctx, p, L, blocks=(a, b))[8]
            assert np.array_equal(acc, want)
    finally:
        ctx.set_option("prefix_kernels", 0)
