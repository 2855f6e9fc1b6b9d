"""HOST parts of the product (no GPU): the O(L) consensus walk, the modal-token sweep, the
writers and the BAM reader — against the golden vectors from the real reference.  The call
records fed to the walk come from the oracle here; the GPU tests feed it the kernel's."""
import json
import os

import numpy as np
import pytest

from oracle import tc_oracle as orc
from tests import synth_small as ss
from trueconsense_amd import Events, Outputs, Sequences, _ffi, engine
from trueconsense_amd.io import bamwriter

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name + ".json")) as fh:
        return json.load(fh)


def gffdict(orfs):
    return {k: {"seqid": "S", "source": "x", "type": "CDS", "start": o["start"], "end": o["end"], "score": ".",
                "strand": o["strand"], "phase": "0", "attributes": "ID=o%d;Name=orf%d" % (k, k),
                "Name": "orf%d" % k, "ID": "o%d" % k} for k, o in enumerate(orfs)}


def test_consensus_walk_matches_reference():
    cases = load("consensus")
    n_ok = n_raise = 0
    for case in cases:
        counts = np.array(case["counts"], dtype=np.int64)
        ins = None if not case["inserts"] else {int(k): v for k, v in case["inserts"].items()}
        for key, exp in case["expected"].items():
            amb, inc = key[3] == "1", key[-1] == "1"
            try:
                plain, alt, flags = orc.call_records(counts, case["mincov"], amb)
            except ZeroDivisionError:
                plain = alt = flags = None
            if "raises" in exp:
                n_raise += 1
                if exp["raises"] == "KeyError":
                    with pytest.raises(engine.WalkKeyError) as ei:
                        Sequences.consensus_from_records(plain, alt, flags, gffdict(case["orfs"]), ins, inc)
                    assert ei.value.args[0] == exp["arg"]
                else:
                    with pytest.raises(ZeroDivisionError):
                        Sequences.consensus_from_records(plain, alt, flags, gffdict(case["orfs"]), ins, inc)
                continue
            cons, gff = Sequences.consensus_from_records(plain, alt, flags, gffdict(case["orfs"]), ins, inc)
            assert cons == exp["consensus"], (case["name"], key)
            assert [[gff[k]["start"], gff[k]["end"]] for k in sorted(gff)] == exp["orfs"], (case["name"], key)
            n_ok += 1
    assert n_ok > 600 and n_raise > 5


def test_modal_tokens_match_oracle_and_reference():
    for case in load("outputs"):
        spec = case["spec"]
        reads = ss.reads_from_spec(spec)
        L = len(case["counts"])
        want = {}
        for p in range(1, L + 1):
            toks = orc.region_tokens(reads, p)
            from collections import Counter
            want[p] = (Counter(t.upper() for t in toks).most_common(1)[0][0] if toks else None, len(toks))
        got = engine.modal_tokens(reads, range(1, L + 1))
        assert got == want, case["name"]
    # Events.ExtractInserts post-pileup logic against the reference's answers
    for case in load("extract"):
        toks = case["tokens"] or []
        from collections import Counter
        modal = Counter(t.upper() for t in toks).most_common(1)[0][0] if toks else None
        assert Events._parse_token(modal) == (case["bases"], case["size"])


class _Hdr:
    raw_text = "##gff-version 3\n"


def test_writers_match_reference(tmp_path):
    for case in load("outputs"):
        spec = case["spec"]
        reads = ss.reads_from_spec(spec)
        counts = np.array(case["counts"], np.int64)
        idict = {i + 1: dict(zip(orc.COLS, (int(v) for v in counts[i]))) for i in range(len(counts))}
        for key, run in case["runs"].items():
            amb = key == "amb1"
            plain, alt, flags = orc.call_records(counts, spec["mincov"], amb)
            has, ins = Events.inserts_from_flags(flags, reads)
            if "raises" in run:
                with pytest.raises((KeyError, ZeroDivisionError, IndexError)):
                    c1, gff = Sequences.consensus_from_records(plain, alt, flags, gffdict(spec["orfs"]), ins, True)
                    c0, _ = Sequences.consensus_from_records(plain, alt, flags, gffdict(spec["orfs"]), ins, False)
                    Outputs.vcf_text("DATE", ["ARGS"], "ref.fa", "refid", list(spec["ref"]), c0, idict,
                                     spec["mincov"], has, ins)
                continue
            c1, gff = Sequences.consensus_from_records(plain, alt, flags, gffdict(spec["orfs"]), ins, True)
            c0, _ = Sequences.consensus_from_records(plain, alt, flags, gffdict(spec["orfs"]), ins, False)
            assert ">SAMPLE mincov=%d\n%s\n" % (spec["mincov"], c1) == run["fa"], case["name"]
            vcf = Outputs.vcf_text("DATE", ["ARGS"], "ref.fa", "refid", list(spec["ref"]), c0, idict,
                                   spec["mincov"], has, ins)
            assert vcf == run["vcf"], case["name"]
            p = tmp_path / "o.gff"
            Outputs.WriteGFF(_Hdr, gff, str(p), "SAMPLE")
            assert p.read_text() == run["gff"], case["name"]
            from trueconsense_amd.Coverage import BuildCoverage
            BuildCoverage(idict, str(p))
            assert p.read_text() == run["tsv"]


def test_bam_writer_reader_roundtrip(tmp_path):
    for case in load("outputs")[:6]:
        reads = ss.reads_from_spec(case["spec"])
        path = str(tmp_path / (case["name"] + ".bam"))
        bamwriter.write_bam(path, reads, "refid", len(case["spec"]["ref"]), block=700)   # many small BGZF blocks
        bam = engine.BamFile(path, threads=3)
        assert bam.references == ("refid",) and bam.lengths == (len(case["spec"]["ref"]),)
        assert bam.n_reads == reads["n_reads"] and bam.n_blocks > 3
        got = bam.arrays()
        for k in ("pos", "flag", "l_qseq", "tid"):
            assert np.array_equal(got[k], reads[k]), k
        n = reads["n_reads"]
        assert np.array_equal(got["cigar_off"], reads["cigar_off"])
        assert np.array_equal(got["cigar"][:int(reads["cigar_off"][n])], reads["cigar"])
        assert np.array_equal(got["seq_off"], reads["seq_off"])
        assert np.array_equal(got["seq"][:int(reads["seq_off"][n])], reads["seq"])
        assert np.array_equal(got["qual"][:len(reads["qual"])], reads["qual"])
        assert engine.reads_extent(bam, 0) == engine.reads_extent(reads, 0)
        assert engine.modal_tokens(bam, [5, 9, 30]) == engine.modal_tokens(reads, [5, 9, 30])


def test_bam_reader_rejects_garbage(tmp_path):
    from trueconsense_amd._ffi import TcmiError
    p = tmp_path / "x.bam"
    p.write_bytes(b"this is not a bam file at all, not even gzip")
    with pytest.raises(TcmiError):
        engine.BamFile(str(p))
    reads = ss.reads_from_spec(load("outputs")[0]["spec"])
    good = tmp_path / "g.bam"
    bamwriter.write_bam(str(good), reads, "refid", 100)
    raw = bytearray(good.read_bytes())
    raw[len(raw) // 2] ^= 0xFF                       # corrupt a deflate byte -> CRC / inflate failure
    p.write_bytes(bytes(raw))
    with pytest.raises(TcmiError):
        engine.BamFile(str(p))
    with pytest.raises(TcmiError):
        engine.BamFile(str(tmp_path / "missing.bam"))


def test_walk_fast_runs_equal_position_by_position():
    """The walk skips over quiet runs in bulk; with that switched off it goes position by position.
    Both must agree on long random inputs (events, inserts of every size, rows of both strands,
    overlapping rows, premature stops), including where they raise."""
    import ctypes as C
    from oracle import c_oracle
    from trueconsense_amd import _ffi
    lib = _ffi.lib()
    lib.tcmi_walk_set_fast_runs.argtypes = [C.c_int]
    lib.tcmi_walk_set_fast_runs.restype = None
    rng = np.random.default_rng(99)
    n_diff_checked = n_raised = 0
    try:
        for rep in range(60):
            L = int(rng.integers(300, 4000))
            seq = rng.integers(0, 4, L)
            cov = int(rng.choice([40, 100, 300]))
            m = np.zeros((L, 7), np.int32)
            m[:, 0] = cov
            m[np.arange(L), 1 + np.array([0, 1, 2, 3])[seq]] = cov      # columns A,T,C,G
            mincov = int(rng.choice([1, 30, 50]))
            ins_pos, ins_shift, ins_seqs = [], [], []
            for _ in range(int(rng.integers(0, 25))):
                p = int(rng.integers(1, L - 12))
                kind = int(rng.integers(0, 7))
                if kind == 0:                                            # deletion run
                    n = int(rng.integers(1, 9))
                    x = int(cov * float(rng.choice([0.95, 0.6])))
                    for q in range(p, min(L - 1, p + n)):
                        row = m[q - 1].copy(); b = int(np.argmax(row[1:5])) + 1
                        m[q - 1, 1:] = 0; m[q - 1, b] = cov - x; m[q - 1, 5] = x
                elif kind == 1:                                          # minority deletion + run
                    x = int(cov * 0.2)
                    b = int(np.argmax(m[p - 1, 1:5])) + 1
                    m[p - 1, 1:] = 0; m[p - 1, b] = cov - x; m[p - 1, 5] = x
                    for q in range(p + 1, min(L - 1, p + 1 + int(rng.integers(1, 5)))):
                        b = int(np.argmax(m[q - 1, 1:5])) + 1
                        m[q - 1, 1:] = 0; m[q - 1, b] = cov // 10; m[q - 1, 5] = cov - cov // 10
                elif kind == 2:                                          # low / zero coverage
                    for q in range(p, min(L, p + int(rng.integers(1, 30)))):
                        m[q - 1] = 0
                        m[q - 1, 0] = int(rng.choice([0, mincov - 1 if mincov > 1 else 0]))
                elif kind == 3 and p not in ins_pos:                      # insert
                    ins_pos.append(p)
                    ins = "".join("ACGT"[int(k)] for k in rng.integers(0, 4, int(rng.choice([1, 2, 3, 12]))))
                    ins_seqs.append(ins)
                    ins_shift.append(int(str(len(ins))[-1]))
                    m[p - 1, 6] = cov
                elif kind == 4:                                          # stop codon TAA
                    for j, col in enumerate((2, 1, 1)):
                        m[p - 1 + j, 1:5] = 0
                        m[p - 1 + j, col] = cov
                elif kind == 5:                                          # ambiguity / lower case
                    b = int(np.argmax(m[p - 1, 1:5])) + 1
                    m[p - 1, 1:5] = 0; m[p - 1, b] = cov // 2; m[p - 1, 1 + (b % 4)] = cov - cov // 2
            order = np.argsort(ins_pos)
            ins_pos = [ins_pos[i] for i in order]; ins_shift = [ins_shift[i] for i in order]; ins_seqs = [ins_seqs[i] for i in order]
            orfs = []
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(1, L - 30))
                orfs.append((a, min(L, a + 3 * int(rng.integers(3, 300))), bool(rng.random() < 0.8)))
            plain, alt, flags = c_oracle.call(m, mincov, bool(rng.random() < 0.5))
            res = []
            for fast in (1, 0):
                lib.tcmi_walk_set_fast_runs(fast)
                for inc in (True, False):
                    try:
                        c, ns, ne = engine.consensus_walk(plain, alt, flags, [o[0] for o in orfs], [o[1] for o in orfs],
                                                          [o[2] for o in orfs], ins_pos, ins_shift, ins_seqs, inc)
                        res.append((c, ns.tolist(), ne.tolist()))
                    except (engine.WalkKeyError, ZeroDivisionError) as e:
                        res.append((type(e).__name__, e.args))
                        n_raised += 1
            assert res[0] == res[2] and res[1] == res[3], rep
            n_diff_checked += 1
    finally:
        lib.tcmi_walk_set_fast_runs(1)
    assert n_diff_checked == 60


def test_bam_reader_survives_damaged_files(tmp_path):
    """Truncations and bit flips (in the BGZF framing, in the deflate stream, and in the inflated BAM
    after re-compression): the reader either decodes or raises TcmiError — it never crashes or hangs."""
    import struct
    import zlib
    from trueconsense_amd._ffi import TcmiError
    reads = ss.reads_from_spec(load("outputs")[2]["spec"])
    good = tmp_path / "g.bam"
    bamwriter.write_bam(str(good), reads, "refid", 150, block=900)
    raw = good.read_bytes()
    rng = np.random.default_rng(5)
    p = tmp_path / "d.bam"
    n_ok = n_err = 0
    for trial in range(120):
        data = bytearray(raw)
        kind = trial % 3
        if kind == 0:
            data = data[:int(rng.integers(1, len(data)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                data[int(rng.integers(0, len(data)))] ^= 1 << int(rng.integers(0, 8))
        else:
            # damage the *inflated* stream and re-wrap it as one valid BGZF member (CRC matches): exercises the BAM parser
            blocks, o = [], 0
            while o < len(raw):
                bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
                blocks.append(zlib.decompress(raw[o + 18:o + bsize - 8], -15))
                o += bsize
            inflated = bytearray(b"".join(blocks))
            for _ in range(int(rng.integers(1, 6))):
                inflated[int(rng.integers(0, len(inflated)))] = int(rng.integers(0, 256))
            if rng.random() < 0.5:
                inflated = inflated[:int(rng.integers(4, len(inflated)))]
            data = bytearray()
            for k in range(0, len(inflated), 60000):
                data += bamwriter._bgzf_block(bytes(inflated[k:k + 60000]), 1)
        p.write_bytes(bytes(data))
        try:
            bam = engine.BamFile(str(p), threads=2)
            a = bam.arrays()
            assert len(a["pos"]) == bam.n_reads
            engine.reads_extent(bam, 150)
            n_ok += 1
        except TcmiError:
            n_err += 1
    assert n_ok + n_err == 120 and n_err > 40


def test_windowed_token_sweep_equals_full_sweep(tmp_path):
    """With the sorted / max-span promise the sweep visits only the reads around each candidate; the result
    must equal the full sweep (and the BAM reader must compute a correct promise)."""
    from tests import fuzz_reads as fz
    rng = np.random.default_rng(12)
    for rep in range(6):
        L = 3000
        reads = fz.random_reads(rng, 1500, L, long_reads=(rep % 2 == 1))
        positions = sorted(set(int(x) for x in rng.integers(1, L, 60)))
        if rep < 4:
            reads["tid"][:] = 0                      # (unplaced reads in between make a BAM unsorted: last two reps)
        full = engine.modal_tokens(reads, positions)
        path = str(tmp_path / "w.bam")
        bamwriter.write_bam(path, reads, "refid", L)
        bam = engine.BamFile(path, threads=2)
        st, _ = bam.as_struct()
        spans = [sum(l for op, l in orc.read_cigar(reads, i) if op in (0, 2, 3, 7, 8)) for i in range(reads["n_reads"])]
        if rep < 4:
            assert bam.sorted == 1 and st.sorted_max_span == max(spans)
        else:
            assert bam.sorted == 0 and st.sorted_max_span == 0
        assert engine.modal_tokens(bam, positions) == full
        if rep >= 4:
            continue
        hinted = dict(reads)
        hinted["sorted_max_span"] = max(spans)
        assert engine.modal_tokens(hinted, positions) == full


def test_insert_tokens_max_depth_and_overlapping_mates():
    """The two remaining defaults of pysam's pileup behind Events.ExtractInserts (Events.py:66): max_depth = 8000 and
    ignore_overlaps.  The native sweep against the oracle's independent restatement (which tweaks the whole overlap of a
    pair, as htslib does, where the product only evaluates the column)."""
    rng = np.random.default_rng(8)
    # ---- a 9 500-deep insert column: the first reads carry the insertion, the later ones (dropped by the cap) do not ----
    reads = []
    for k in range(9500):
        carrier = k < 4200
        reads.append({"pos": 100, "flag": 0, "cigar": "10M2I10M" if carrier else "20M",
                      "seq": "ACGTACGTAC" + ("GG" if carrier else "") + "ACGTACGTAC", "qual": 30})
    reads += [{"pos": 101 + int(j), "flag": 16, "cigar": "12M", "seq": "CGTACGTACACG", "qual": 35} for j in range(300) for _ in range(1)]
    reads.sort(key=lambda r: r["pos"])
    deep = ss.reads_from_spec({"reads": reads})
    deep["sorted_max_span"] = 20
    for max_depth, expect_modal in ((8000, "C+2GG"), (0, "C")):
        want = orc.region_tokens(deep, 110, max_depth=max_depth)
        got = engine.modal_tokens(deep, [110], max_depth=max_depth)[110]
        assert got[1] == len(want) and got[0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0] == expect_modal
    assert len(orc.region_tokens(deep, 110)) == 8000 + 9            # 8 000 admitted at the shared start, then every later start
    # ---- overlapping mates: agreeing bases, differing bases, low qualities that only pass when summed, orphans ----
    pairs = []
    for k in range(400):
        start = 200 + int(rng.integers(0, 8))
        mate = start + int(rng.integers(0, 10))
        base1 = "ACGT"[int(rng.integers(0, 4))]
        base2 = base1 if rng.random() < 0.6 else "ACGT"[int(rng.integers(0, 4))]
        q1, q2 = int(rng.integers(5, 41)), int(rng.integers(5, 41))
        seq1 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30))
        seq2 = seq1[mate - start:] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, mate - start))
        # the column of interest is 1-based 215 (0-based 214)
        i1, i2 = 214 - start, 214 - mate
        seq1 = seq1[:i1] + base1 + seq1[i1 + 1:]
        seq2 = seq2[:i2] + base2 + seq2[i2 + 1:]
        ins = rng.random() < 0.7
        cig1 = "%dM1I%dM" % (i1 + 1, 30 - i1 - 2) if ins else "30M"
        flags = (99, 147) if rng.random() < 0.9 else (65, 129)        # proper pair / paired but not proper (orphans are filtered)
        name = "pair%d" % k
        pairs.append({"pos": start, "flag": flags[0], "cigar": cig1, "seq": seq1, "qual": [q1] * 30, "name": name, "mtid": 0, "mpos": mate,
                      "tlen": mate + 30 - start})
        pairs.append({"pos": mate, "flag": flags[1], "cigar": "30M", "seq": seq2, "qual": [q2] * 30, "name": name, "mtid": 0, "mpos": start,
                      "tlen": -(mate + 30 - start)})
    pairs.sort(key=lambda r: r["pos"])
    pr = ss.reads_from_spec({"reads": pairs})
    pr["sorted_max_span"] = 30
    for col in (215, 214, 216):
        for olap in (True, False):
            want = orc.region_tokens(pr, col, ignore_overlaps=olap)
            got = engine.modal_tokens(pr, [col], ignore_overlaps=olap)[col]
            assert got[1] == len(want), (col, olap, got[1], len(want))
            assert got[0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0]
    assert len(orc.region_tokens(pr, 215)) < len(orc.region_tokens(pr, 215, ignore_overlaps=False))
    # ---- a mate with a deletion / ref-skip ON the column: its token is tested on the quality of its NEXT query base, which the
    #      tweak reaches on that base's reference position if the other mate has a matched base there (the product probes the
    #      other mate; the oracle tweaks the whole overlap) ----
    one = ss.reads_from_spec({"reads": [
        {"pos": 10, "flag": 99, "cigar": "5M2D10M", "seq": "ACGTACGTACGTACG", "qual": 30, "name": "p", "mtid": 0, "mpos": 12, "tlen": 30},
        {"pos": 12, "flag": 147, "cigar": "15M", "seq": "ACGTACGTACGTACG", "qual": 30, "name": "p", "mtid": 0, "mpos": 10, "tlen": -30}]})
    want = orc.region_tokens(one, 16)
    got = engine.modal_tokens(one, [16])[16]
    assert got[1] == len(want) and (not want or got[0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0])
    assert engine.modal_tokens(one, [16], ignore_overlaps=False)[16][1] == 2
    dels = []
    for k in range(600):
        start = 300 + int(rng.integers(0, 6))
        mate = start + int(rng.integers(0, 8))
        seq1 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 40))
        seq2 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 40))
        q1 = [int(x) for x in rng.integers(4, 30, 40)]
        q2 = [int(x) for x in rng.integers(4, 30, 40)]
        # column of interest: 1-based 320 (0-based 319); a deletion / skip of 1-4 bases that covers it in one mate or both, sometimes
        # followed by an insertion (then the next query base has no reference position) or preceded by one
        def cigar_with_gap(read_start, seq_len):
            i = 319 - read_start                              # matched bases before the column
            gap = int(rng.integers(1, 5))
            lead = int(rng.integers(0, gap))                  # the gap starts `lead` positions before the column
            kind = "N" if rng.random() < 0.2 else "D"
            a = i - lead
            if rng.random() < 0.2:
                return "%dM%d%s2I%dM" % (a, gap, kind, seq_len - a - 2)
            if rng.random() < 0.15 and a > 3:
                return "%dM1I%dM%d%s%dM" % (a - 2, 1, gap, kind, seq_len - a)
            return "%dM%d%s%dM" % (a, gap, kind, seq_len - a)
        which = rng.random()
        cig1 = cigar_with_gap(start, 40) if which < 0.7 else "40M"
        cig2 = cigar_with_gap(mate, 40) if which > 0.5 else "40M"
        if rng.random() < 0.5:                                # agreeing bases at the sites more often than chance
            seq2 = seq1[mate - start:] + seq2[:mate - start]
        name = "d%d" % k
        proper = rng.random() < 0.9
        dels.append({"pos": start, "flag": 99 if proper else 65, "cigar": cig1, "seq": seq1, "qual": q1, "name": name, "mtid": 0, "mpos": mate,
                     "tlen": mate + 44 - start})
        dels.append({"pos": mate, "flag": 147 if proper else 129, "cigar": cig2, "seq": seq2, "qual": q2, "name": name, "mtid": 0, "mpos": start,
                     "tlen": -(mate + 44 - start)})
    dels.sort(key=lambda r: r["pos"])
    dr = ss.reads_from_spec({"reads": dels})
    dr["sorted_max_span"] = 50
    seen_change = False
    for col in (320, 319, 321, 322):
        want = orc.region_tokens(dr, col)
        got = engine.modal_tokens(dr, [col])[col]
        assert got[1] == len(want), (col, got[1], len(want))
        assert got[0] == orc.Counter(t.upper() for t in want).most_common(1)[0][0]
        seen_change |= len(want) != len(orc.region_tokens(dr, col, ignore_overlaps=False))
    assert seen_change


def test_index_dict_behaves_like_the_reference_dictionary():
    """_state.IndexDict stands for IndexDF.to_dict("index") (TrueConsense.py:237) but makes its 30 000 row dictionaries on demand."""
    from trueconsense_amd import _state
    from trueconsense_amd.Coverage import BuildCoverage, GetCoverage
    c = np.arange(70, dtype=np.int32).reshape(10, 7)
    d = _state.IndexDict(c)
    ref = {i + 1: dict(zip(_state.COLS, row)) for i, row in enumerate(c.tolist())}
    assert len(d) == 10 and list(d)[:3] == [1, 2, 3] and 10 in d and 0 not in d and 11 not in d
    assert d[3] == ref[3] and d.get(11) is None and d.get(4)["A"] == ref[4]["A"]
    with pytest.raises(KeyError):
        d[11]
    with pytest.raises(KeyError):
        GetCoverage(d, 0)
    assert GetCoverage(d, 10) == 63 == GetCoverage(ref, 10)
    assert d == ref and dict(d.items()) == ref and list(d.values())[0] == ref[1]
    assert np.array_equal(_state.counts_of(d), c) and np.array_equal(_state.counts_of(ref), c)
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        BuildCoverage(d, os.path.join(t, "a.tsv"))
        BuildCoverage(ref, os.path.join(t, "b.tsv"))
        assert open(os.path.join(t, "a.tsv")).read() == open(os.path.join(t, "b.tsv")).read() == "".join("%d\t%d\n" % (i + 1, 7 * i) for i in range(10))
