"""HOST parts of the product (no GPU): the O(L) consensus walk, the modal-token sweep, the
writers and the BAM reader — against the golden vectors from the real reference.  The call
records fed to the walk come from the oracle here; the GPU tests feed it the kernel's."""
import json
import os

import numpy as np
import pytest

from oracle import tc_oracle as orc
from tests import synth_small as ss
from trueconsense_amd import Events, Outputs, Sequences, engine
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
