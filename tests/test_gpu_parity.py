"""GPU parity (run with -m gpu on an MI355X): the HIP tally / call kernels and the whole
BAM -> FASTA/VCF/GFF/TSV path, through the C ABI, against the golden vectors from the real
reference and against the oracle on seeded inputs.  Integer / byte results: bit-exact."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from oracle import c_oracle
from oracle import tc_oracle as orc
from tests import synth_small as ss
from trueconsense_amd import Events, Sequences, _ffi, _state, engine, indexing
from trueconsense_amd import synthetic as sy
from trueconsense_amd.io import bamwriter

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name + ".json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def ctx():
    c = _state.default_context()          # raises loudly without a GPU / without libtcmi.so
    yield c


def gffdict(orfs):
    return {k: {"seqid": "S", "source": "x", "type": "CDS", "start": o["start"], "end": o["end"], "score": ".",
                "strand": o["strand"], "phase": "0", "attributes": "ID=o%d;Name=orf%d" % (k, k)}
            for k, o in enumerate(orfs)}


# ------------------------------------------------------------------ stage A
def test_tally_matches_reference_counts(ctx):
    for case in load("outputs"):
        reads = ss.reads_from_spec(case["spec"])
        got = ctx.tally(reads, ref_len=len(case["spec"]["ref"]))
        assert got.tolist() == case["counts"], case["name"]


def _check_tally(ctx, reads, L):
    want = c_oracle.tally(reads, L)
    got = ctx.tally(reads, L=L)
    assert np.array_equal(got, want)
    return got


def test_tally_seeded_mid_size_with_indels(ctx):
    ref, orfs = sy.make_reference()
    reads = sy.make_reads(ref, 200_000, seed=11, indel_sites=sy.default_indel_sites(orfs))
    got = _check_tally(ctx, reads, len(ref))
    assert got[:, 5].sum() > 1000 and got[:, 6].sum() > 1000


def test_tally_edge_cases(ctx):
    ref, _ = sy.make_reference(L=5000, cds=[(10, 600)])
    base = sy.make_reads(ref, 3000, seed=3)
    # sparse reads: every workgroup window misses most of its reads
    sparse = sy.make_reads(ref, 40, seed=4)
    _check_tally(ctx, sparse, len(ref))
    # unsorted input: still correct (window misses fall back to global atomics)
    perm = np.random.default_rng(0).permutation(base["n_reads"])
    shuf = dict(base)
    shuf["pos"], shuf["flag"] = base["pos"][perm], base["flag"][perm]
    shuf["seq"] = base["seq"].reshape(base["n_reads"], -1)[perm].reshape(-1)
    _check_tally(ctx, shuf, len(ref))
    # long reads (span > LDS window) and reads running past the reference end
    spec = {"reads": [{"pos": 10, "flag": 0, "cigar": "900M", "seq": "ACGT" * 225},
                      {"pos": 12, "flag": 16, "cigar": "5M700N5M", "seq": "ACGTACGTAC"},
                      {"pos": 4990, "flag": 0, "cigar": "30M", "seq": "ACGTAC" * 5},
                      {"pos": 100, "flag": 0, "cigar": "4M2D3M1I2M", "seq": "ACGTACGTAC"},
                      {"pos": 100, "flag": 4, "cigar": "10M", "seq": "ACGTACGTAC"},
                      {"pos": 2000, "flag": 0, "cigar": "10M", "seq": "*"},
                      {"pos": 300, "flag": 0, "cigar": "3M", "seq": "NRA"}]}
    reads = ss.reads_from_spec(spec)
    L = engine.reads_extent(reads, len(ref))
    assert L == 5020
    got = _check_tally(ctx, reads, L)
    assert got[2005, 0] == 1 and got[2005, 1:].sum() == 0           # SEQ '*' -> N tokens: coverage only
    # nothing piles up / no reads at all
    none = ss.reads_from_spec({"reads": [{"pos": 5, "flag": 4, "cigar": "10M", "seq": "ACGTACGTAC"}]})
    assert ctx.tally(none, L=50).sum() == 0
    empty = ss.reads_from_spec({"reads": []})
    assert ctx.tally(empty, L=50).sum() == 0
    with pytest.raises(_ffi.TcmiError):
        ctx.tally(base, L=100)                                       # L smaller than the read extent


def test_both_tally_kernels_agree(ctx):
    """aligned reads take the fast kernel by default; option tally_variant=1 forces every read
    through the CIGAR-walk kernel: both must give the oracle's matrix."""
    import ctypes as C2
    ref, orfs = sy.make_reference(L=9000, cds=[(50, 4000), (4200, 8800)])
    reads = sy.make_reads(ref, 60_000, seed=21, indel_sites=sy.default_indel_sites(orfs))
    # soft / hard clips and =/X runs are still "aligned"; odd clip lengths exercise the odd-offset packer
    extra = ss.reads_from_spec({"reads": [
        {"pos": 1000, "flag": 0, "cigar": "3S20M2S", "seq": "NNNACGTACGTACGTACGTACGTAA"},
        {"pos": 1001, "flag": 16, "cigar": "2H4S10=1X9M", "seq": "ACGTACGTACGTACGTACGTACGT"},
        {"pos": 1002, "flag": 0, "cigar": "20M", "seq": "ACGTNACGTRACGTACG=ACG"[:20]},
        {"pos": 1003, "flag": 0, "cigar": "30M", "seq": "ACGTACGTAC"},              # SEQ shorter than the CIGAR
        {"pos": 1004, "flag": 0, "cigar": "700M", "seq": "ACGT" * 175}]})             # too long for the fast kernel
    L = len(ref)
    want = c_oracle.tally(reads, L)
    want_x = c_oracle.tally(extra, L)
    for variant, project in ((0, 1), (0, 0), (1, 1)):
        ctx.set_option("tally_variant", variant)
        ctx.set_option("project_reads", project)
        rs = ctx.upload(reads)
        a, c, g = (C2.c_int64(0) for _ in range(3))
        _ffi.check(_ffi.lib().tcmi_readset_sets(rs.handle, C2.byref(a), C2.byref(c), C2.byref(g)))
        if variant == 1:
            assert a.value == 0 and g.value == rs.n_piled
        elif project:
            assert g.value == 0 and a.value >= rs.n_piled        # indel reads are projected onto the reference
        else:
            assert a.value > 50_000 and g.value > 500 and a.value + g.value == rs.n_piled and c.value >= a.value // 4096
        rs.free()
        assert np.array_equal(ctx.tally(reads, L=L), want), (variant, project)
        assert np.array_equal(ctx.tally(extra, L=L), want_x), (variant, project)
    ctx.set_option("project_reads", 1)
    ctx.set_option("tally_variant", 0)
    rs = ctx.upload(extra)
    a, c, g = (C2.c_int64(0) for _ in range(3))
    _ffi.check(_ffi.lib().tcmi_readset_sets(rs.handle, C2.byref(a), C2.byref(c), C2.byref(g)))
    assert (a.value, g.value) == (6, 0)                              # the 700M read is cut into two pieces
    rs.free()


def test_device_packer_against_host_packer_and_oracle(ctx):
    """The default upload copies the BAM-native arrays to the device and packs them with HIP kernels (CIGAR projection,
    token classes, coverage runs, 4-bit -> bit planes); option device_pack = 0 packs on the host.  Both must give the
    oracle's matrix, and the read sets must say who packed them: the device takes what is sorted or not as long as no
    entry exceeds 512 positions; long reads and far positions go to the host packer."""
    from tests import fuzz_reads as fz
    rng = np.random.default_rng(99)
    ref, orfs = sy.make_reference()
    cases = [("150M + indel carriers", sy.make_reads(ref, 150_000, seed=41, indel_sites=sy.default_indel_sites(orfs)), len(ref), True),
             ("every CIGAR op, odd SEQ content", fz.random_reads(rng, 20_000, 3000, long_reads=False, sort=True), None, True),
             ("unsorted (chunk budget runs out: host packer)", fz.random_reads(rng, 3000, 1500, long_reads=False, sort=False), None, False),
             ("long reads", fz.random_reads(rng, 2000, 20_000, long_reads=True, sort=True), None, False),
             ("mixed lengths 1..512", _uniform_reads(rng, np.sort(rng.integers(0, 5000, 20_000)).astype(np.int32), rng.integers(1, 513, 20_000)), 5600, True),
             ("60k reads on one start", _uniform_reads(rng, np.full(60_000, 77, np.int32), np.full(60_000, 150)), 300, True),
             ("513-base reads", _uniform_reads(rng, np.sort(rng.integers(0, 900, 500)).astype(np.int32), np.full(500, 513)), 1500, False)]
    seen = {}
    try:
        for name, reads, L, on_device in cases:
            L = L or engine.reads_extent(reads, 0)
            want = c_oracle.tally(reads, L)
            for dp in (1, 0):
                ctx.set_option("device_pack", dp)
                rs = ctx.upload(reads)
                assert rs.packed_on_device == bool(dp and on_device), (name, dp)
                got = ctx.step(rs, L, 30, True)[3]
                assert np.array_equal(got, want), (name, dp, np.argwhere(got != want)[:5])
                seen.setdefault(name, []).append((rs.n_piled, rs.algorithmic_bytes, rs.max_end))
                rs.free()
    finally:
        ctx.set_option("device_pack", 1)
    assert all(a == b for a, b in seen.values()), seen              # kept reads, algorithmic bytes, extent: same from both packers


def test_tally_accumulate_split_readsets(ctx):
    """cfg 5 shape: one BAM split into contiguous read ranges, partial matrices summed."""
    ref, _ = sy.make_reference(L=8000, cds=[(10, 900)])
    reads = sy.make_reads(ref, 30_000, seed=9)
    L, ld = len(ref), 8192
    whole = c_oracle.tally(reads, L)
    import torch
    d_counts = torch.zeros(7 * ld, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                    # torch's stream is not the context's stream
    n = reads["n_reads"]
    nb = len(reads["seq"]) // n
    for part, (a, b) in enumerate(((0, n // 3), (n // 3, n))):
        sub = {"n_reads": b - a, "pos": reads["pos"][a:b], "flag": reads["flag"][a:b], "l_qseq": reads["l_qseq"][a:b],
               "cigar_off": reads["cigar_off"][a:b + 1] - reads["cigar_off"][a], "cigar": reads["cigar"][a:b],
               "seq_off": reads["seq_off"][a:b + 1] - reads["seq_off"][a], "seq": reads["seq"][a * nb:b * nb]}
        rs = ctx.upload(sub)
        _ffi.check(_ffi.lib().tcmi_tally_dev(ctx.handle, rs.handle, L, ld, C.c_void_p(d_counts.data_ptr()), int(part == 0)),
                   ctx.handle)
        ctx.sync()
        rs.free()
    got = d_counts.cpu().numpy().reshape(7, ld)[:, :L].T
    assert np.array_equal(got, whole)


def test_full_size_1m_reads_exact_and_deterministic(ctx):
    """BASELINE config 2: 29 903 bp x 1M reads; compared with the C oracle and run twice."""
    ref, _ = sy.make_reference()
    reads = sy.make_reads(ref, 1_000_000, seed=2)
    L = len(ref)
    rs = ctx.upload(reads)
    assert rs.packed_on_device
    assert rs.n_piled == 1_000_000 and rs.algorithmic_bytes == 91 * 1_000_000
    p1, a1, f1, c1 = ctx.step(rs, L, 30, True)
    p2, a2, f2, c2 = ctx.step(rs, L, 30, True)
    assert np.array_equal(c1, c2) and np.array_equal(p1, p2) and np.array_equal(f1, f2)
    assert int(c1[:, 0].sum()) == 150_000_000                       # every base is one token
    assert np.array_equal(c1[:, 1:6].sum(1), c1[:, 0])              # ACGT-only reads: classes partition coverage
    want = c_oracle.tally(reads, L)
    assert np.array_equal(c1, want)
    wp, wa, wf = c_oracle.call(want, 30, True)
    assert np.array_equal(p1, wp) and np.array_equal(a1, wa) and np.array_equal(f1, wf)
    # steps that leave the counts on the device: the call kernel zeroes the matrix behind itself
    for rep in range(5):
        p3, a3, f3, _ = ctx.step(rs, L, 30, True, want_counts=False)
        assert np.array_equal(p3, wp) and np.array_equal(a3, wa) and np.array_equal(f3, wf), rep
    assert np.array_equal(ctx.step(rs, L, 30, True)[3], want)
    rs.free()


def _reads_slice(reads, a, b):
    """Reads [a, b) of a dict of flat arrays as a dict of its own (offsets rebased)."""
    co, so, qo = (np.asarray(reads[k]) for k in ("cigar_off", "seq_off", "qual_off"))
    out = {"n_reads": b - a}
    for k in ("pos", "flag", "l_qseq", "tid"):
        out[k] = reads[k][a:b]
    out["cigar_off"], out["cigar"] = (co[a:b + 1] - co[a]).astype(np.uint64), reads["cigar"][int(co[a]):int(co[b])]
    out["seq_off"], out["seq"] = (so[a:b + 1] - so[a]).astype(np.uint64), reads["seq"][int(so[a]):int(so[b])]
    out["qual_off"], out["qual"] = (qo[a:b + 1] - qo[a]).astype(np.uint64), reads["qual"][int(qo[a]):int(qo[b])]
    return out


def _oracle_tokens_at(reads, pos1):
    """orc.region_tokens on the (sorted) reads that can reach column pos1 - 1: the emulator loops over reads in Python."""
    c, span = pos1 - 1, int(reads["sorted_max_span"])
    a = int(np.searchsorted(reads["pos"], c - span + 1, "left"))
    b = int(np.searchsorted(reads["pos"], c, "right"))
    return orc.region_tokens(_reads_slice(reads, a, b), pos1)


def test_full_size_indel_workload_configs2(ctx, tmp_path):
    """BASELINE configs[2] at its full size: 1M reads, indel carriers at the CDS boundaries.  Counts against the scalar C
    oracle; accepted inserts and both consensus walks against the oracle chain; the same through a BAM FILE decoded on the
    device."""
    ref, orfs = sy.make_reference()
    L = len(ref)
    reads = sy.make_reads(ref, 1_000_000, seed=12, indel_sites=sy.default_indel_sites(orfs))
    want = c_oracle.tally(reads, L)
    rs = ctx.upload(reads)
    assert rs.packed_on_device
    plain, alt, flags, counts = ctx.step(rs, L, 30, True)
    rs.free()
    assert np.array_equal(counts, want)
    assert counts[:, 5].sum() > 20_000 and counts[:, 6].sum() > 20_000
    wp, wa, wf = c_oracle.call(want, 30, True)
    assert np.array_equal(plain, wp) and np.array_equal(alt, wa) and np.array_equal(flags, wf)
    # candidates -> modal tokens: the native sweep over the 1M reads against the oracle's emulated region pileup
    cand = Events.candidates_from_flags(flags)
    assert len(cand) >= 3
    has, ins = Events.inserts_from_flags(flags, reads)
    ohas, oins = orc.list_inserts(want, 30, lambda p: _oracle_tokens_at(reads, p))
    assert has and ins == oins and len(ins) >= 3
    gff = {k: {"start": o["start"], "end": o["end"], "strand": o["strand"]} for k, o in enumerate(orfs)}
    for inc in (True, False):
        wcons, worfs = orc.build_consensus(30, want.astype(np.int64), [dict(o) for o in orfs], True, oins, inc)
        got, ggff = Sequences.consensus_from_records(plain, alt, flags, gff, ins, inc)
        assert got == wcons, inc
        assert [[ggff[k]["start"], ggff[k]["end"]] for k in sorted(ggff)] == [[o["start"], o["end"]] for o in worfs]
    # the same reads as a BAM file: inflated, indexed and packed on the device, insert tokens from the lazily decoded file
    p = str(tmp_path / "cfg2.bam")
    bamwriter.write_bam(p, reads, "MN908947.3", L, level=1)
    runner = engine.FileRunner(ctx, [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs], 30)
    text = runner.run([p], names=["S"], ref_len=L)[0]
    wcons = orc.build_consensus(30, want.astype(np.int64), [dict(o) for o in orfs], True, oins, True)[0]
    assert text == orc.fasta_text("S", 30, wcons) and runner.decoded_on == {"device": 1, "host": 0}
    runner.close()


def test_full_size_split_bam_tile_configs4(ctx):
    """The per-GPU share of BASELINE configs[4]: 6.25M reads confined to one eighth of the genome (~250 000x there).
    255 reads per lane and chunk, the balanced chunk size and the coverage runs all see their deepest input."""
    ref, _ = sy.make_reference()
    L = len(ref)
    tile = (L - 150 + 1) // 8
    reads = sy.make_reads(ref, 6_250_000, seed=13, start_range=(3 * tile, 4 * tile))
    want = c_oracle.tally(reads, L)
    rs = ctx.upload(reads)
    assert rs.packed_on_device and rs.n_piled == 6_250_000
    counts = ctx.step(rs, L, 30, True)[3]
    rs.free()
    assert int(counts[:, 0].sum()) == 150 * 6_250_000
    assert np.array_equal(counts, want)
    assert counts[:, 0].max() > 200_000
    ctx.set_option("device_pack", 0)                             # and the host packer on the same input
    try:
        rs = ctx.upload(reads)
        assert not rs.packed_on_device
        assert np.array_equal(ctx.step(rs, L, 30, True)[3], want)
        rs.free()
    finally:
        ctx.set_option("device_pack", 1)


# ------------------------------------------------------------------ stage B
def test_call_matches_reference_rows(ctx):
    rows = load("rows")
    m = np.array([c["row"] for c in rows], np.int32)
    for mincov in (1, 30):
        for amb in (True, False):
            plain, alt, flags, ev = ctx.call(m, mincov, amb, want_events=True)
            wp, wa, wf = c_oracle.call(m, mincov, amb)
            assert np.array_equal(plain, wp) and np.array_equal(alt, wa) and np.array_equal(flags, wf)
            assert ev.tolist() == np.nonzero(flags & 14)[0].tolist()
            for i, case in enumerate(rows):
                rk = case["rank"]
                assert bool(flags[i] & _ffi.F_PRIMX) == (rk[0][0] == "X")
                assert bool(flags[i] & _ffi.F_AMBIG) == case["ambig"][0]
                if case["mindel"] != "ZeroDivisionError":
                    assert bool(flags[i] & _ffi.F_MINDEL) == case["mindel"]
                else:
                    assert flags[i] & _ffi.F_COVZERO
                assert bool(flags[i] & _ffi.F_INSCAND) == case["inscand"][str(mincov)]
                if amb and case["ambig"][0] and not (flags[i] & _ffi.F_LOWCOV):
                    assert chr(plain[i]) == case["ambig"][1]


def test_call_random_matrix_against_oracle(ctx):
    rng = np.random.default_rng(5)
    L = 70_001
    cov = rng.integers(0, 400, L)
    m = np.zeros((L, 7), np.int32)
    frac = rng.dirichlet([0.6] * 6, L)
    m[:, 1:6] = np.floor(frac[:, :5] * cov[:, None]).astype(np.int32)
    m[:, 0] = cov
    m[:, 6] = (rng.random(L) < 0.3) * rng.integers(0, 400, L)
    m[:, 6] = np.minimum(m[:, 6], m[:, 0])
    for mincov, amb in ((30, True), (0, False), (100, True)):
        got = ctx.call(m, mincov, amb)
        want = c_oracle.call(m, mincov, amb)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)


def test_build_consensus_matches_reference(ctx):
    class Bam:
        def __init__(self, region):
            self.region = region

        def modal_token(self, p):
            from collections import Counter
            toks = self.region.get(p)
            return Counter(t.upper() for t in toks).most_common(1)[0][0] if toks else None

    n_ok = n_raise = 0
    for case in load("consensus"):
        counts = np.array(case["counts"], np.int32)
        idict = {i + 1: dict(zip(orc.COLS, (int(v) for v in counts[i]))) for i in range(len(counts))}
        bam = Bam({int(k): v for k, v in case["region"].items()})
        has, ins = Events.ListInserts(idict, case["mincov"], bam)
        assert ins == (None if not case["inserts"] else {int(k): v for k, v in case["inserts"].items()})
        for key, exp in case["expected"].items():
            amb, inc = key[3] == "1", key[-1] == "1"
            if "raises" in exp:
                n_raise += 1
                with pytest.raises((KeyError, ZeroDivisionError)) as ei:
                    Sequences.BuildConsensus(case["mincov"], idict, gffdict(case["orfs"]), amb, bam, inc)
                assert type(ei.value).__name__ in (exp["raises"], "WalkKeyError")
                continue
            cons, gff = Sequences.BuildConsensus(case["mincov"], idict, gffdict(case["orfs"]), amb, bam, inc)
            assert cons == exp["consensus"], (case["name"], key)
            assert [[gff[k]["start"], gff[k]["end"]] for k in sorted(gff)] == exp["orfs"]
            n_ok += 1
    assert n_ok > 600 and n_raise > 5


# ------------------------------------------------------------------ whole path through files
def test_cli_bam_to_outputs_matches_reference(ctx, tmp_path, monkeypatch):
    from trueconsense_amd import TrueConsense as cli
    monkeypatch.chdir(tmp_path)
    n_done = n_over = 0
    for case in load("outputs"):
        spec = case["spec"]
        reads = ss.reads_from_spec(spec)
        bamwriter.write_bam("in.bam", reads, "refid", len(spec["ref"]))
        with open("ref.fa", "w") as fh:
            fh.write(">refid some description\n")
            for o in range(0, len(spec["ref"]), 60):
                fh.write(spec["ref"][o:o + 60] + "\n")
        with open("f.gff", "w") as fh:
            fh.write("##gff-version 3\n")
            for k, o in enumerate(spec["orfs"]):
                fh.write("S\tx\tCDS\t%d\t%d\t.\t%s\t0\tID=o%d;Name=orf%d\n" % (o["start"], o["end"], o["strand"], k, k))
        df = indexing.BuildIndex("in.bam", "ref.fa")
        assert df.values.tolist() == case["counts"] and list(df.columns) == list(orc.COLS)
        assert list(df.index) == list(range(1, len(case["counts"]) + 1))
        for key, run in case["runs"].items():
            argv = ["-i", "in.bam", "-ref", "ref.fa", "-gff", "f.gff", "-cov", str(spec["mincov"]), "-name", "SAMPLE",
                    "-o", "out.fa", "-vcf", "out.vcf", "-ogff", "out.gff", "-doc", "out.tsv"]
            if key == "amb0":
                argv.append("-noambig")
            monkeypatch.setattr(sys, "argv", ["TrueConsense", "ARGS"])
            if "raises" in run:
                with pytest.raises((KeyError, ZeroDivisionError, IndexError)):
                    cli.main(argv)
                continue
            cli.main(argv)
            assert open("out.fa").read() == run["fa"], case["name"]
            assert open("out.tsv").read() == run["tsv"]
            lines = open("out.vcf").read().split("\n")
            lines[1] = "##fileDate=DATE"
            assert "\n".join(lines) == run["vcf"], case["name"]
            assert open("out.gff").read() == run["gff_cli"], case["name"]      # the whole corrected GFF, every column
            n_done += 1
        # --index-override through the command line: rows of the tallied matrix replaced before the call kernel
        # (TrueConsense.py:232-235, indexing.py:39-72); golden text from the reference's own functions
        orun = case.get("override_run")
        if orun:
            import gzip
            with gzip.open("over.csv.gz", "wt") as fh:
                fh.write(orun["csv"])
            argv = ["-i", "in.bam", "-ref", "ref.fa", "-gff", "f.gff", "-cov", str(spec["mincov"]), "-name", "SAMPLE", "-o", "out.fa",
                    "-vcf", "out.vcf", "-ogff", "out.gff", "-doc", "out.tsv", "--index-override", "over.csv.gz"]
            monkeypatch.setattr(sys, "argv", ["TrueConsense", "ARGS"])
            if "raises" in orun:
                with pytest.raises((KeyError, ZeroDivisionError, IndexError)):
                    cli.main(argv)
            else:
                cli.main(argv)
                assert open("out.fa").read() == orun["fa"] and open("out.tsv").read() == orun["tsv"], case["name"]
                lines = open("out.vcf").read().split("\n")
                lines[1] = "##fileDate=DATE"
                assert "\n".join(lines) == orun["vcf"] and open("out.gff").read() == orun["gff"], case["name"]
                n_over += 1
    assert n_done >= 20 and n_over >= 10


def test_cli_batch_writes_the_same_four_files(ctx, tmp_path, monkeypatch):
    """--batch MANIFEST: many samples through the native file runner, whose walker threads write FASTA, VCF, corrected GFF and
    coverage TSV themselves (csrc/pipeline.cpp) — byte for byte the reference's text (the golden files of the single-sample
    command line above), for two manifest lines at a time."""
    from trueconsense_amd import TrueConsense as cli
    monkeypatch.chdir(tmp_path)
    n_done = n_raise = 0
    for case in load("outputs"):
        spec = case["spec"]
        reads = ss.reads_from_spec(spec)
        bamwriter.write_bam("in.bam", reads, "refid", len(spec["ref"]))
        with open("ref.fa", "w") as fh:
            fh.write(">refid some description\n")
            for o in range(0, len(spec["ref"]), 60):
                fh.write(spec["ref"][o:o + 60] + "\n")
        with open("f.gff", "w") as fh:
            fh.write("##gff-version 3\n")
            for k, o in enumerate(spec["orfs"]):
                fh.write("S\tx\tCDS\t%d\t%d\t.\t%s\t0\tID=o%d;Name=orf%d\n" % (o["start"], o["end"], o["strand"], k, k))
        with open("m.tsv", "w") as fh:
            fh.write("# BAM, name, FASTA, VCF, GFF, TSV\n")
            fh.write("in.bam\tSAMPLE\ta.fa\ta.vcf\ta.gff\ta.tsv\n")
            fh.write("in.bam\tSAMPLE\tb.fa\t-\tb.gff\n")                    # (no VCF, no TSV for the second)
        for key, run in case["runs"].items():
            argv = ["--batch", "m.tsv", "-ref", "ref.fa", "-gff", "f.gff", "-cov", str(spec["mincov"])] + (["-noambig"] if key == "amb0" else [])
            monkeypatch.setattr(sys, "argv", ["TrueConsense", "ARGS"])
            for f in ("a.fa", "a.vcf", "a.gff", "a.tsv", "b.fa", "b.gff", "b.vcf"):
                if os.path.exists(f):
                    os.remove(f)
            if "raises" in run:
                with pytest.raises((KeyError, ZeroDivisionError, IndexError)):
                    cli.main(argv)
                n_raise += 1
                continue
            cli.main(argv)
            assert open("a.fa").read() == open("b.fa").read() == run["fa"], case["name"]
            assert open("a.tsv").read() == run["tsv"]
            lines = open("a.vcf").read().split("\n")
            lines[1] = "##fileDate=DATE"
            assert "\n".join(lines) == run["vcf"], case["name"]
            assert open("a.gff").read() == open("b.gff").read() == run["gff_cli"], case["name"]
            assert not os.path.exists("b.vcf")
            n_done += 1
    assert n_done >= 20


def test_cli_gpus_n_batch_shards_and_one_bam_split(ctx, tmp_path, monkeypatch):
    """--gpus N (additive flag; the reference's flag surface is TrueConsense.py:75-209): N child processes started before any GPU
    call.  --batch: the manifest dealt round-robin to N runners (configs[3]); -i: ONE BAM file shared by N ranks, a reduce of the count
    matrix, entries of the insert candidates gathered to rank 0 (configs[4]).  Rehearsed with N = 2 on this one GPU (gloo for the
    exchange): every output file byte-identical to the golden text / to the single-GPU command line."""
    import subprocess
    env = dict(os.environ, TCMI_SPLIT_ONE_GPU="1", TCMI_SPLIT_BACKEND="gloo", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    monkeypatch.chdir(tmp_path)
    case = next(c for c in load("outputs") if "raises" not in c["runs"]["amb1"] and c["runs"]["amb1"]["fa"].count("\n") == 2)
    spec, run = case["spec"], case["runs"]["amb1"]
    bamwriter.write_bam("in.bam", ss.reads_from_spec(spec), "refid", len(spec["ref"]))
    with open("ref.fa", "w") as fh:
        fh.write(">refid some description\n" + spec["ref"] + "\n")
    with open("f.gff", "w") as fh:
        fh.write("##gff-version 3\n")
        for k, o in enumerate(spec["orfs"]):
            fh.write("S\tx\tCDS\t%d\t%d\t.\t%s\t0\tID=o%d;Name=orf%d\n" % (o["start"], o["end"], o["strand"], k, k))
    with open("m.tsv", "w") as fh:
        for k in range(5):
            fh.write("in.bam\tSAMPLE\ts%d.fa\ts%d.vcf\ts%d.gff\ts%d.tsv\n" % (k, k, k, k))
    base = [sys.executable, "-m", "trueconsense_amd.TrueConsense", "-ref", "ref.fa", "-gff", "f.gff", "-cov", str(spec["mincov"])]
    r = subprocess.run(base + ["--batch=m.tsv", "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    for k in range(5):
        assert open("s%d.fa" % k).read() == run["fa"] and open("s%d.tsv" % k).read() == run["tsv"]
        assert open("s%d.gff" % k).read() == run["gff_cli"]
        lines, want = open("s%d.vcf" % k).read().split("\n"), run["vcf"].split("\n")
        assert lines[:1] + lines[3:] == want[:1] + want[3:]         # (all but the date and the ##source line, which carries each child's own argv)
    # ONE file over two ranks, indel carriers with candidate columns in the middle of the file: the four outputs = the single-GPU command line's
    ref, orfs = sy.make_reference(L=4000, cds=[(100, 1900), (2100, 3900)])
    sites = [(1900, "I", "A", 0.9), (1990, "I", "GT", 0.7), (2005, "I", "ACGTACGTACGTAC", 0.8), (2010, "D", 3, 0.6), (3000, "I", "TT", 0.95)]
    reads = sy.make_reads(ref, 6000, seed=91, indel_sites=sites)
    bamwriter.write_bam("one.bam", reads, "MN", len(ref), block=3000, split_records=True)
    with open("r2.fa", "w") as fh:
        fh.write(">MN x\n" + ref + "\n")
    head, body = sy.gff_text(orfs, seqid="MN")
    with open("g2.gff", "w") as fh:
        fh.write(head + body)
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2"])):
        argv = [sys.executable, "-m", "trueconsense_amd.TrueConsense", "-i", "one.bam", "-ref", "r2.fa", "-gff", "g2.gff", "-cov", "30", "-name", "S",
                "-o", tag + ".fa", "-vcf", tag + ".vcf", "-ogff", tag + ".gff", "-doc", tag + ".tsv"] + extra
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        vcf = open(tag + ".vcf").read().split("\n")
        outs[tag] = (open(tag + ".fa").read(), open(tag + ".gff").read(), open(tag + ".tsv").read(), vcf[:1] + vcf[3:])
    assert outs["one"] == outs["two"]
    # ... and the split worker itself with the exchange it uses on a node of GPUs: the C hook over an RCCL communicator (one rank here)
    argv = [sys.executable, "-m", "trueconsense_amd.split_main", "-i", "one.bam", "-ref", "r2.fa", "-gff", "g2.gff", "-cov", "30", "-name", "S",
            "-o", "hook.fa", "-vcf", "hook.vcf", "-ogff", "hook.gff", "-doc", "hook.tsv", "--gpus", "1"]
    env1 = {k: v for k, v in env.items() if k not in ("TCMI_SPLIT_ONE_GPU", "TCMI_SPLIT_BACKEND")}
    r = subprocess.run(argv, env=dict(env1, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TCMI_SPLIT_VERBOSE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "tcmi_rccl_reduce: ncclReduce" in r.stderr, r.stderr[-1500:]
    vcf = open("hook.vcf").read().split("\n")
    assert (open("hook.fa").read(), open("hook.gff").read(), open("hook.tsv").read(), vcf[:1] + vcf[3:]) == outs["one"]
    assert "ACGTACGTACGTAC" in outs["two"][0]                        # (the 14-base insertion, whose bases travelled as text, is in the consensus)


def test_two_streams_per_context_give_the_same_outputs(tmp_path, monkeypatch):
    """TCMI_STREAM_SPLIT (the scheduling experiment of round 6: everything behind the inflate on a second stream per context — by
    priority, on compute units of its own, or both; all three lose to one stream and stay behind the variable): the code
    path must stay right.  `--batch` through the native runner in a process of its own per mode: four outputs per sample = the golden text."""
    import subprocess
    case = next(c for c in load("outputs") if "raises" not in c["runs"]["amb1"] and c["runs"]["amb1"]["fa"].count("\n") == 2)
    spec, run = case["spec"], case["runs"]["amb1"]
    monkeypatch.chdir(tmp_path)
    bamwriter.write_bam("in.bam", ss.reads_from_spec(spec), "refid", len(spec["ref"]))
    with open("ref.fa", "w") as fh:
        fh.write(">refid some description\n" + spec["ref"] + "\n")
    with open("f.gff", "w") as fh:
        fh.write("##gff-version 3\n")
        for k, o in enumerate(spec["orfs"]):
            fh.write("S\tx\tCDS\t%d\t%d\t.\t%s\t0\tID=o%d;Name=orf%d\n" % (o["start"], o["end"], o["strand"], k, k))
    for mode in ("1", "2", "3"):
        with open("m.tsv", "w") as fh:
            for k in range(6):
                fh.write("in.bam\tSAMPLE\tm%s_%d.fa\tm%s_%d.vcf\tm%s_%d.gff\tm%s_%d.tsv\n" % ((mode, k) * 4))
        env = dict(os.environ, TCMI_STREAM_SPLIT=mode, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-m", "trueconsense_amd.TrueConsense", "--batch", "m.tsv", "-ref", "ref.fa", "-gff", "f.gff", "-cov", str(spec["mincov"])],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (mode, r.stderr[-1500:])
        for k in range(6):
            assert open("m%s_%d.fa" % (mode, k)).read() == run["fa"] and open("m%s_%d.tsv" % (mode, k)).read() == run["tsv"], mode
            assert open("m%s_%d.gff" % (mode, k)).read() == run["gff_cli"], mode


def test_configs1_full_size_from_bam_files_through_eight_contexts(ctx, tmp_path):
    """BASELINE configs[1] at full size FROM BAM FILES through the runner the bench's headline uses (FileRunner.run_resident, eight
    GPU contexts, the files' compressed bytes resident in HBM): every FASTA text against the Python oracle chain on its file
    (oracle/bam_oracle.c + tally_oracle.c + tc_oracle.py list_inserts / build_consensus — no product code in the checker)."""
    import bench
    ref, orfs = sy.make_reference()
    L = len(ref)
    paths = []
    for k in range(2):
        n = 1_000_000
        reads = sy.make_reads(ref, n, seed=8100 + k)
        p = str(tmp_path / ("full%d.bam" % k))
        bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)
        paths.append(p)
        del reads
    rows = [{"start": o["start"], "end": o["end"], "strand": o["strand"]} for o in orfs]
    runner = engine.FileRunner(ctx, rows, 30, True, decoders=2, decode_threads=4, walkers=2, gpu_streams=8)
    dbams = [engine.DeviceBam(p).to_device(ctx) for p in paths]
    n_items = 24
    texts = runner.run_resident([dbams[i % 2] for i in range(n_items)], names=["S%d" % (i % 2) for i in range(n_items)], ref_len=L)
    assert runner.decoded_on == {"device": n_items, "host": 0}
    want = [r[0] for r in bench.oracle_chain_many([(p, [dict(o) for o in orfs], L, 30, "S%d" % k) for k, p in enumerate(paths)], 2)]
    for i, t in enumerate(texts):
        assert t == want[i % 2], i
    assert len(texts[0].split("\n")[1]) == L
    for d in dbams:
        d.close()
    runner.close()


def test_self_cleaning_steps_and_pipeline(ctx):
    """Steps that do not fetch the counts leave the matrix zeroed by the call kernel (no memset
    between them); the native pipeline must give what the step-by-step path gives, inserts included."""
    from trueconsense_amd.engine import Pipeline
    ref, orfs = sy.make_reference(L=6000, cds=[(100, 2500), (2600, 5800)])
    L = len(ref)
    sites = [(500, "I", "ACG", 0.9), (1200, "D", 3, 0.95), (3000, "D", 1, 0.9), (4000, "I", "T", 0.6)]
    sets = [sy.make_reads(ref, 30_000, seed=31, indel_sites=sites),
            sy.make_reads(ref, 25_000, seed=32)]
    pipe = Pipeline(0, slots=3, walkers=2)
    pipe.set_orfs([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    rss = [pipe.ctx.upload(r) for r in sets]
    want = []
    for reads, rs in zip(sets, rss):
        # three steps on one workspace: no counts, no counts, counts -> the last must still be exact
        pipe.ctx.step(rs, L, 30, True, want_counts=False)
        p1, a1, f1, _ = pipe.ctx.step(rs, L, 30, True, want_counts=False)
        p2, a2, f2, counts = pipe.ctx.step(rs, L, 30, True, want_counts=True)
        assert np.array_equal(counts, c_oracle.tally(reads, L))
        assert np.array_equal(p1, p2) and np.array_equal(f1, f2) and np.array_equal(a1, a2)
        has, ins = Events.inserts_from_flags(f2, reads)
        gff = {k: {"start": o["start"], "end": o["end"], "strand": "+"} for k, o in enumerate(orfs)}
        want.append(Sequences.consensus_from_records(p2, a2, f2, gff, ins, True)[0])
    assert "-" in want[0] and len(want[0]) > L                     # deletions and spliced inserts are present
    order = [0, 1, 1, 0, 0, 1, 0]
    out, status = pipe.run([rss[i] for i in order], L, 30, True, host_reads=[sets[i] for i in order])
    assert not status.any()
    assert [o.decode() for o in out] == [want[i] for i in order]
    with pytest.raises(_ffi.TcmiError):                             # insert candidates but no host reads
        pipe.run([rss[0]], L, 30, True)
    assert pipe.last_status[0] == _ffi.E_UNSUPPORTED
    out, _ = pipe.run([rss[1]], L, 30, True)                        # no candidates: fine without host reads
    assert out[0].decode() == want[1]
    # the ride-along call must also work when a step has no fast-kernel launch to carry it: a read set that
    # takes the CIGAR-walk kernel only, and one without reads (its consensus is all N)
    pipe.ctx.set_option("tally_variant", 1)
    rs_general = pipe.ctx.upload(sets[1])
    pipe.ctx.set_option("tally_variant", 0)
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and k not in ("cigar_off", "seq_off", "qual_off") else v) for k, v in sets[1].items()}
    empty.update(n_reads=0, cigar_off=np.zeros(1, np.uint64), seq_off=np.zeros(1, np.uint64), qual_off=np.zeros(1, np.uint64))
    rs_empty = pipe.ctx.upload(empty)
    p0, a0, f0, _ = pipe.ctx.step(rs_empty, L, 30, True, want_counts=False)
    gff = {k: {"start": o["start"], "end": o["end"], "strand": "+"} for k, o in enumerate(orfs)}
    want_empty = Sequences.consensus_from_records(p0, a0, f0, gff, {}, True)[0]
    assert set(want_empty) == {"N"} and len(want_empty) == L
    items = [rss[1], rs_general, rs_empty, rss[1], rs_empty, rs_general, rs_general, rss[1]]
    exp = [want[1], want[1], want_empty, want[1], want_empty, want[1], want[1], want[1]]
    for defer in (1, 0):
        pipe.ctx.set_option("defer_call", defer)
        out, status = pipe.run(items, L, 30, True)
        assert not status.any() and [o.decode() for o in out] == exp, defer
    pipe.ctx.set_option("defer_call", 1)
    pipe.close()


def test_fuzz_random_cigars_all_paths(ctx):
    """20k random reads with every CIGAR op, odd SEQ content and flags: the fast kernel with reads
    projected onto the reference, the fast + CIGAR-walk split, and the CIGAR-walk kernel alone must
    all reproduce the oracle, sorted or not, short windows or long reads."""
    from tests import fuzz_reads as fz
    rng = np.random.default_rng(20251121)
    for rep, (n, L, long_reads, sort) in enumerate(((20000, 3000, False, True), (4000, 20000, True, True),
                                                   (3000, 1500, False, False), (500, 100000, True, True))):
        reads = fz.random_reads(rng, n, L, long_reads=long_reads, sort=sort)
        Lx = engine.reads_extent(reads, L)
        want = c_oracle.tally(reads, Lx)
        for variant, project in ((0, 1), (0, 0), (1, 1)):
            ctx.set_option("tally_variant", variant)
            ctx.set_option("project_reads", project)
            got = ctx.tally(reads, L=Lx)
            assert np.array_equal(got, want), (rep, variant, project, np.argwhere(got != want)[:5])
    ctx.set_option("tally_variant", 0)
    ctx.set_option("project_reads", 1)


def test_long_reference_sparse_reads(ctx):
    """A 5 Mb reference with a few thousand scattered reads: one-read chunks, a large matrix, the call
    kernel and the host walk over millions of positions."""
    from tests import fuzz_reads as fz
    rng = np.random.default_rng(7)
    L = 5_000_000
    reads = fz.random_reads(rng, 3000, L, long_reads=True)
    want = c_oracle.tally(reads, L)
    rs = ctx.upload(reads)
    plain, alt, flags, counts = ctx.step(rs, L, 1, True)
    rs.free()
    assert np.array_equal(counts, want)
    wp, wa, wf = c_oracle.call(want, 1, True)
    assert np.array_equal(plain, wp) and np.array_equal(alt, wa) and np.array_equal(flags, wf)
    cons, ns, ne = engine.consensus_walk(plain, alt, flags, [10, 2_000_000], [1_000_000, 4_999_990], [1, 1], [], [], [], True)
    assert len(cons) == L and cons.count("-") > 0


def test_batched_readset_and_pipeline(ctx):
    """Several BAMs in one read set at shifted positions: one tally launch and one call launch for the
    batch; every BAM's slice must equal what it gives on its own, through ctx.step and the pipeline."""
    from trueconsense_amd.engine import Pipeline
    ref, orfs = sy.make_reference(L=5000, cds=[(100, 2400), (2600, 4800)])
    L, stride = len(ref), 5120
    sites = [(700, "I", "GT", 0.8), (1500, "D", 3, 0.9), (3000, "D", 2, 0.6)]
    bams = [sy.make_reads(ref, 12_000 + 1000 * k, seed=50 + k, indel_sites=sites if k % 2 == 0 else None) for k in range(5)]
    pipe = Pipeline(0, slots=2, walkers=3)
    pipe.set_orfs([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    single = [pipe.ctx.upload(r) for r in bams]
    want_counts = [c_oracle.tally(r, L) for r in bams]
    want_cons, _ = pipe.run(single, L, 30, True, host_reads=bams)
    # batch of 3 + batch of 2 (different sizes: two read sets, run separately)
    for group in ([0, 1, 2], [3, 4]):
        rs = pipe.ctx.upload_batch([bams[i] for i in group], stride)
        plain, alt, flags, counts = pipe.ctx.step(rs, len(group) * stride, 30, True)
        for k, i in enumerate(group):
            assert np.array_equal(counts[k * stride:k * stride + L], want_counts[i]), i
            assert not counts[k * stride + L:(k + 1) * stride].any()
        # the pipeline attaches the call of step k to the tally launch of step k + 1 (option defer_call, default);
        # with it off every step is a tally launch and a call launch
        for defer in (1, 0, 1):
            pipe.ctx.set_option("defer_call", defer)
            out, status = pipe.run([rs, rs, rs], L, 30, True, host_reads=[bams[i] for i in group] * 3, batch=len(group),
                                   pos_stride=stride)
            assert not status.any()
            assert out == [want_cons[i] for i in group] * 3, defer
        rs.free()
    with pytest.raises(_ffi.TcmiError):
        pipe.ctx.upload_batch(bams[:2], 1024)                       # stride smaller than the reads' extent
    pipe.close()


def test_mid_size_consensus_against_the_oracle_chain(ctx):
    """BAM-shaped reads -> HIP tally/call -> native inserts + walk, against the oracle's whole chain
    (emulated pileup -> reference-pinned list_inserts / build_consensus) on 30k reads with accepted
    inserts, deletions inside and outside ORFs, with and without ambiguity codes."""
    ref, orfs = sy.make_reference(L=3000, cds=[(100, 1300), (1500, 2800)])
    L = len(ref)
    sites = [(400, "I", "ACG", 0.95), (700, "D", 3, 0.95), (1400, "D", 2, 0.9), (2000, "I", "T", 0.9),
             (2300, "D", 1, 0.3), (2500, "I", "GGTTACGTACGT", 0.92)]
    reads = sy.make_reads(ref, 30_000, seed=77, indel_sites=sites)
    counts = ctx.tally(reads, L=L)
    ocounts = orc.tally_matrix({k: v for k, v in reads.items()}, L)
    assert np.array_equal(counts, ocounts)
    gff = {k: {"start": o["start"], "end": o["end"], "strand": o["strand"]} for k, o in enumerate(orfs)}
    olist = [dict(o) for o in orfs]
    for amb in (True, False):
        has, ins = orc.list_inserts(ocounts, 30, lambda p: orc.region_tokens(reads, p))
        assert has and len(ins) == 3
        plain, alt, flags = ctx.call(counts, 30, amb)
        h2, ins2 = Events.inserts_from_flags(flags, reads)
        assert ins2 == ins
        for inc in (True, False):
            want, worfs = orc.build_consensus(30, ocounts, olist, amb, ins, inc)
            got, ggff = Sequences.consensus_from_records(plain, alt, flags, gff, ins2, inc)
            assert got == want, (amb, inc)
            assert [[ggff[k]["start"], ggff[k]["end"]] for k in sorted(ggff)] == [[o["start"], o["end"]] for o in worfs]
        assert len(want) == L and "-" in want


def test_api_misuse_is_reported(ctx):
    """Bad arguments come back as TCMI_E_* with a message, not as a crash."""
    ref, _ = sy.make_reference(L=2000, cds=[(10, 900)])
    reads = sy.make_reads(ref, 500, seed=1)
    rs = ctx.upload(reads)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.step(rs, 100, 30, True)                                  # L smaller than the reads' extent
    assert e.value.code == _ffi.E_ARG and "extent" in str(e.value)
    ctx.step_begin(rs, 2000, 30, True, want_counts=False)
    with pytest.raises(_ffi.TcmiError) as e:
        ctx.step_begin(rs, 2000, 30, True, want_counts=False)        # second begin without an end
    assert "not been ended" in str(e.value)
    plain, _, flags, _ = ctx.step_end()
    assert len(plain) == 2000
    with pytest.raises(_ffi.TcmiError):
        _ffi.check(_ffi.lib().tcmi_step_end(ctx.handle, None, None, None, None, None), ctx.handle)   # end without begin
    with pytest.raises(_ffi.TcmiError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(_ffi.TcmiError):
        ctx.call(np.zeros((0, 7), np.int32), 30, True)                # L must be positive
    bad = dict(reads)
    bad["seq"] = reads["seq"][:10]                                    # SEQ shorter than the offsets say
    bad["seq_off"] = reads["seq_off"].copy()
    bad["seq_off"][1:] = 10
    bad["seq_off"][0] = 8
    with pytest.raises(_ffi.TcmiError):
        ctx.upload(bad)
    rs.free()


def test_long_reads_take_the_fast_kernel_in_pieces(ctx):
    """Nanopore-like reads (5-15 kb, dozens of small indels each): projected, cut into 512-position pieces,
    re-sorted by position and tallied by the fast kernel; nothing is left for the CIGAR-walk kernel."""
    import ctypes as C2
    rng = np.random.default_rng(31)
    L = 40_000
    specs = []
    for _ in range(400):
        ops, span, qlen = [], 0, 0
        target = int(rng.integers(5000, 15000))
        while span < target:
            m = int(rng.integers(20, 400))
            ops.append("%dM" % m); span += m; qlen += m
            r = rng.random()
            if r < 0.4:
                k = int(rng.integers(1, 6)); ops.append("%dD" % k); span += k
            elif r < 0.8:
                k = int(rng.integers(1, 6)); ops.append("%dI" % k); qlen += k
        ops.append("30M"); span += 30; qlen += 30
        pos = int(rng.integers(0, L - span))
        seq = "".join("ACGTN"[int(c)] for c in rng.choice(5, qlen, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        specs.append({"pos": pos, "flag": int(rng.choice([0, 16])), "cigar": "".join(ops), "seq": seq})
    specs.sort(key=lambda r: r["pos"])
    reads = ss.reads_from_spec({"reads": specs})
    want = c_oracle.tally(reads, L)
    rs = ctx.upload(reads)
    a, c, g = (C2.c_int64(0) for _ in range(3))
    _ffi.check(_ffi.lib().tcmi_readset_sets(rs.handle, C2.byref(a), C2.byref(c), C2.byref(g)))
    assert g.value == 0 and a.value > 4000                              # ~ span / 512 pieces per read
    plain, alt, flags, counts = ctx.step(rs, L, 5, True)
    rs.free()
    assert np.array_equal(counts, want)
    assert counts[:, 5].sum() > 1000 and counts[:, 6].sum() > 1000


def _uniform_reads(rng, starts, lengths, alphabet="ACGTN"):
    """Flat arrays for reads with CIGAR <len>M at the given starts (already sorted), random bases."""
    n = len(starts)
    lengths = np.asarray(lengths, np.int64)
    nb = (lengths + 1) // 2
    seq_off = np.zeros(n + 1, np.uint64); seq_off[1:] = np.cumsum(nb)
    codes = np.array([1, 2, 4, 8, 15], np.uint8)[:len(alphabet)]
    seq = np.zeros(int(seq_off[-1]), np.uint8)
    for i in range(n):
        c = codes[rng.integers(0, len(codes), int(lengths[i]) + (int(lengths[i]) & 1))]
        if lengths[i] & 1:
            c[-1] = 0
        seq[int(seq_off[i]):int(seq_off[i + 1])] = (c[0::2] << 4) | c[1::2]
    return {"n_reads": n, "pos": np.asarray(starts, np.int32), "flag": np.zeros(n, np.uint16),
            "l_qseq": lengths.astype(np.int32), "tid": np.zeros(n, np.int32),
            "cigar_off": np.arange(n + 1, dtype=np.uint64), "cigar": (lengths.astype(np.uint32) << 4),
            "seq_off": seq_off, "seq": seq, "qual": np.full(int(lengths.sum()), 30, np.uint8),
            "qual_off": np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)}


def test_chunk_geometry_extremes(ctx):
    """Shapes that push the chunker and the lane mapping of the bit-plane kernel to their limits: read lengths around
    the 8- and 32-position word sizes, one-base reads, 600-base reads, tens of thousands of reads on one start
    (narrow window, many depth slices, the 255-reads-per-lane bound), windows at the maximal width, reads that
    start at position 0 and end on the last position."""
    rng = np.random.default_rng(555)
    cases = []
    for length in (1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 150, 599, 600):
        starts = np.sort(rng.integers(0, 40, 3000)).astype(np.int32)
        cases.append(("len %d" % length, _uniform_reads(rng, starts, np.full(3000, length)), 40 + length))
    cases.append(("60k reads on one start", _uniform_reads(rng, np.full(60_000, 77, np.int32), np.full(60_000, 150)), 300))
    cases.append(("40k one-base reads on one position", _uniform_reads(rng, np.full(40_000, 5, np.int32), np.full(40_000, 1)), 10))
    starts = np.sort(rng.integers(0, 5000, 20_000)).astype(np.int32)
    cases.append(("mixed lengths 1..600", _uniform_reads(rng, starts, rng.integers(1, 601, 20_000)), 5600))
    starts = np.sort(rng.integers(0, 200_000, 3000)).astype(np.int32)
    cases.append(("sparse, every window at its widest", _uniform_reads(rng, starts, rng.integers(300, 601, 3000)), 200_600))
    for name, reads, L in cases:
        want = c_oracle.tally(reads, L)
        got = ctx.tally(reads, L=L)
        assert np.array_equal(got, want), (name, np.argwhere(got != want)[:5])
        assert int(got[:, 0].sum()) == int(reads["l_qseq"].sum()), name


def test_pipeline_long_queue_every_item_right(ctx):
    """A few hundred steps through the native pipeline (more items than workspaces, fewer walkers than jobs, the
    ride-along call on and off, batched and single items, with and without accepted inserts): every consensus of the
    queue must be the one its read set gives on its own."""
    from trueconsense_amd.engine import Pipeline
    ref, orfs = sy.make_reference(L=4000, cds=[(100, 1900), (2100, 3900)])
    L, stride = len(ref), 4096
    sites = [(600, "I", "ACG", 0.9), (1500, "D", 2, 0.9), (2500, "I", "T", 0.7)]
    bams = [sy.make_reads(ref, 6000 + 500 * k, seed=900 + k, indel_sites=sites if k % 3 == 0 else None) for k in range(6)]
    for slots, walkers in ((5, 3), (2, 6)):
        pipe = Pipeline(0, slots=slots, walkers=walkers)
        pipe.set_orfs([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
        single = [pipe.ctx.upload(r) for r in bams]
        want, _ = pipe.run(single, L, 30, True, host_reads=bams)
        assert len(set(want)) >= 4 and any(len(w) > L for w in want)            # inserts were spliced in
        rng = np.random.default_rng(slots)
        order = [int(x) for x in rng.integers(0, 6, 240)]
        for defer in (1, 0):
            pipe.ctx.set_option("defer_call", defer)
            out, status = pipe.run([single[i] for i in order], L, 30, True, host_reads=[bams[i] for i in order])
            assert not status.any()
            assert out == [want[i] for i in order], (slots, defer)
        pairs = [(0, 1, 2), (3, 4, 5), (5, 0, 3)]
        rss = [pipe.ctx.upload_batch([bams[i] for i in g], stride) for g in pairs]
        order = [int(x) for x in rng.integers(0, 3, 90)]
        pipe.ctx.set_option("defer_call", 1)
        out, status = pipe.run([rss[j] for j in order], L, 30, True, host_reads=[bams[i] for j in order for i in pairs[j]],
                               batch=3, pos_stride=stride)
        assert not status.any()
        assert out == [want[i] for j in order for i in pairs[j]], slots
        for r in rss + single:
            r.free()
        pipe.close()


@pytest.mark.gpu
def test_configs0_python_standin_beside_the_command_line(tmp_path):
    """BASELINE configs[0] (10k synthetic 150-bp reads over the 29 903-bp reference): the Python stand-in for the reference's whole run
    (TrueConsense.py:212-264 — oracle/tc_oracle.py end to end: per-token tally loop, ListInserts with region pile-ups, both BuildConsensus
    walks, the writers) beside the product's command line on the same BAM file: FASTA text, VCF records, corrected ORF coordinates and
    coverage TSV must be the same.  (bench.py's `configs0` leg, at its own size.)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ref, orfs = sy.make_reference()
    r = bench.configs0_leg(np, sy, ref, orfs, len(ref), str(tmp_path), 30)
    assert "error" not in r, r
    assert r["outputs_equal"] == {"fa": True, "tsv": True, "vcf": True, "gff": True}, r
    assert r["accepted_inserts"] >= 1 and r["vcf_records"] > 10 and r["python_standin_seconds"] > r["product_cli_in_process_seconds"]["tally"]
