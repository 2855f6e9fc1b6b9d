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
