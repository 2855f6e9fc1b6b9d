#!/usr/bin/env python3
"""
make_golden.py — generate the golden vectors under tests/golden/ by running the REAL
reference (imported from /root/reference, never copied) on seeded inputs.

Runs only in the build container (the reference does not exist on the GPU box); the
JSON files it writes are the fixtures that travel.  Usage:

    python tests/golden/make_golden.py            # rewrites tests/golden/*.json

What is exercised: the reference's own modules Ambig, Coverage, Events,
ORFs, Sequences import as they are; indexing / Outputs / TrueConsense import once stub
`pysam`, `AminoExtract` and `Bio` modules sit in sys.modules.  The stub pysam plays back
pileup columns produced by oracle.tc_oracle.pileup_columns (the BAM-bytes -> tokens
step is the one part of the path that cannot be pinned here: pysam is not installable).
"""
import gzip
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TC_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import tc_oracle as orc          # noqa: E402
from tests import synth_small as ss          # noqa: E402

# ----------------------------------------------------------------------------- stubs
_REG = {"bam": {}, "fasta": {}}              # path -> payload


class _Column:
    def __init__(self, pos, toks):
        self.pos = pos
        self._t = toks

    def get_query_sequences(self, add_indels=False, **kw):
        assert add_indels is True
        return list(self._t)


class _Alignment:
    """pysam.AlignmentFile stand-in: plays back precomputed columns."""

    def __init__(self, path, mode="rb"):
        self._p = _REG["bam"][path]
        self.references = ("refid",)
        self.calls = []

    def pileup(self, *args, **kw):
        self.calls.append((args, dict(kw)))
        if not args:                          # stage A (indexing.py:100)
            assert kw == {"stepper": "nofilter", "max_depth": 10000000, "min_base_quality": 0}, kw
            for pos in sorted(self._p["stageA"]):
                yield _Column(pos, self._p["stageA"][pos])
            return
        rname, start, end = args              # Events.py:66
        assert kw == {"truncate": True}
        toks = self._p["region"].get(end)
        if toks is None:
            return
        yield _Column(start, toks)


class _Fasta:
    def __init__(self, path):
        self.lengths = [len(_REG["fasta"][path][1])]


def _install_stubs():
    pysam = types.ModuleType("pysam")
    pysam.AlignmentFile = _Alignment
    pysam.FastaFile = _Fasta
    sys.modules["pysam"] = pysam

    amino = types.ModuleType("AminoExtract")
    amino.SequenceReader = object
    amino.GFFDataFrame = object
    gd = types.ModuleType("AminoExtract.gff_data")

    class GFFColumns:
        @staticmethod
        def get_names():
            return ["seqid", "source", "type", "start", "end", "score", "strand", "phase",
                    "attributes"]
    gd.GFFColumns = GFFColumns
    amino.gff_data = gd
    sys.modules["AminoExtract"] = amino
    sys.modules["AminoExtract.gff_data"] = gd

    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")

    class _Rec:
        def __init__(self, i, s):
            self.id, self.seq = i, s

    def parse(path, fmt):
        assert fmt == "fasta"
        rid, seq = _REG["fasta"][path]
        yield _Rec(rid, seq)
    seqio.parse = parse
    bio.SeqIO = seqio
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio


_install_stubs()
sys.path.insert(0, REF)
from TrueConsense import Ambig, Coverage, Events, ORFs, Outputs, Sequences, indexing  # noqa: E402


# ----------------------------------------------------------------------------- helpers
def idict_from_counts(counts):
    return {i + 1: dict(zip(orc.COLS, (int(v) for v in counts[i]))) for i in range(len(counts))}


def gffdict_from_orfs(orfs, extra=False):
    d = {}
    for k, o in enumerate(orfs):
        row = {"seqid": "S", "source": "x", "type": "CDS", "start": o["start"], "end": o["end"],
               "score": ".", "strand": o["strand"], "phase": "0",
               "attributes": "ID=o%d;Name=orf%d" % (k, k)}
        if extra:
            row["Name"] = "orf%d" % k
            row["ID"] = "o%d" % k
        d[k] = row
    return d


def gffdict_cli(orfs):
    d = {}
    for k, o in enumerate(orfs):
        d[k] = {"seqid": "SAMPLE", "source": "x", "type": "CDS", "start": o["start"], "end": o["end"], "score": ".",
                "strand": o["strand"], "phase": "0", "attributes": "ID=o%d;Name=orf%d" % (k, k), "ID": "o%d" % k, "Name": "orf%d" % k}
    return d


class _TokBam:
    """duck-typed bam for Events.ExtractInserts (Events.py:63-67)."""
    references = ("refid",)

    def __init__(self, region):
        self.region = region

    def pileup(self, rname, start, end, truncate=True):
        toks = self.region.get(end)
        if toks is None:
            return
        yield _Column(start, toks)


def run_consensus(mincov, counts, orfs, include_ambig, region, include_ins):
    try:
        cons, gff = Sequences.BuildConsensus(mincov, idict_from_counts(counts),
                                             gffdict_from_orfs(orfs), include_ambig,
                                             _TokBam(region), include_ins)
        return {"consensus": cons,
                "orfs": [[gff[k]["start"], gff[k]["end"]] for k in sorted(gff)]}
    except (KeyError, ZeroDivisionError) as e:
        return {"raises": type(e).__name__,
                "arg": (int(e.args[0]) if isinstance(e, KeyError) else None)}


# ----------------------------------------------------------------------------- sections
def golden_tokens(rng):
    """indexing.py:102-132 through the unmodified BuildIndex (stub pysam playback)."""
    alphabet = ["A", "a", "C", "c", "G", "g", "T", "t", "N", "n", "*", ">", "<", ".", ",",
                "A+2TT", "*+1A", "C-2NN", "g+1a", "t-1n", "*", "R", "y", ">+1C", "=", "M+3ACG"]
    cases = [["A", "a", "*", "A+2TT", "*+1A", "C-2NN", ">", "N", "g", "T"], []]
    for _ in range(40):
        n = int(rng.integers(1, 60))
        cases.append([alphabet[int(k)] for k in rng.integers(0, len(alphabet), n)])
    out = []
    for k, toks in enumerate(cases):
        if not toks:
            continue
        path = "tok%d.bam" % k
        _REG["bam"][path] = {"stageA": {4: toks}, "region": {}}
        _REG["fasta"]["r.fa"] = ("refid", "A" * 8)
        df = indexing.BuildIndex(path, "r.fa")
        d = df.to_dict("index")
        assert sorted(d) == list(range(1, 9))
        out.append({"tokens": toks, "row": [int(d[5][c]) for c in orc.COLS],
                    "empty_row": [int(d[1][c]) for c in orc.COLS]})
    return out


def golden_rows(rng):
    """Sequences.GetNucleotide, Ambig.IsAmbiguous, Events.MinorityDel, insert threshold."""
    rows = [[0] * 7, [10, 5, 5, 0, 0, 0, 0], [136, 45, 50, 0, 41, 0, 0],
            [100, 55, 0, 45, 0, 0, 0], [20, 11, 0, 9, 0, 0, 0], [10, 6, 0, 4, 0, 0, 0],
            [100, 0, 0, 0, 0, 0, 55], [20, 0, 0, 0, 0, 0, 11], [100, 80, 0, 0, 0, 15, 0],
            [20, 17, 0, 0, 0, 3, 0], [60, 51, 0, 0, 0, 9, 0], [1000, 851, 0, 0, 0, 149, 0],
            [100, 25, 25, 25, 25, 0, 0], [99, 33, 33, 33, 0, 0, 0], [100, 34, 33, 33, 0, 0, 0],
            [100, 30, 30, 0, 0, 40, 0], [100, 30, 30, 0, 0, 30, 0], [100, 40, 0, 0, 30, 30, 0],
            [90, 30, 30, 30, 0, 0, 0], [100, 20, 20, 20, 20, 20, 0]]
    for _ in range(600):
        cov = int(rng.integers(1, 400))
        kind = int(rng.integers(0, 5))
        if kind == 0:                         # one dominant base
            v = rng.multinomial(cov, [0.9, 0.04, 0.03, 0.02, 0.01])
        elif kind == 1:                       # near-ties around the 10-point window
            v = rng.multinomial(cov, [0.45, 0.35, 0.15, 0.04, 0.01])
        elif kind == 2:
            v = rng.multinomial(cov, [0.3, 0.28, 0.26, 0.1, 0.06])
        elif kind == 3:
            v = rng.multinomial(cov, [0.25, 0.25, 0.24, 0.22, 0.04])
        else:
            v = rng.multinomial(cov, [0.2] * 5)
        v = [int(x) for x in rng.permutation(v)]
        other = int(rng.integers(0, 4))       # N / refskip tokens only count as coverage
        ins = int(rng.integers(0, cov + 1)) if rng.random() < 0.3 else 0
        rows.append([cov + other] + v + [ins])
    out = []
    for r in rows:
        idict = {1: dict(zip(orc.COLS, r))}
        nucs = [list(Sequences.GetNucleotide(idict, 1, k)) for k in (1, 2, 3, 4, 5)]
        amb = Ambig.IsAmbiguous(*[tuple(n) for n in nucs[:4]], r[0])
        try:
            md = bool(Events.MinorityDel(idict, 1))
        except ZeroDivisionError:
            md = "ZeroDivisionError"
        cand = {}
        for mincov in (1, 30):
            seen = []

            class B:
                references = ("r",)

                def pileup(self, *a, **k):
                    seen.append(a)
                    return iter(())
            Events.ListInserts(idict, mincov, B())
            cand[str(mincov)] = bool(seen)
        out.append({"row": r, "rank": [[n, int(c)] for n, c in nucs],
                    "ambig": [bool(amb[0]), amb[1]], "mindel": md, "inscand": cand})
    return out


def golden_extract(rng):
    """Events.ExtractInserts post-pileup logic (Events.py:68-82)."""
    cases = [["C+12ACGTACGTACGT"] * 55 + ["c+12acgtacgtacgt"] * 5 + ["C"] * 40,
             ["A-2NN"] * 3 + ["A+1T"] * 2, ["A", "C", "A"], [], ["A+1T", "A+1G"],
             ["a+1g", "A+1T", "A+1G", "a+1t"], ["*+2AC"] * 3 + ["A"], ["*"] * 4,
             ["G+3ACN", "G+3ACN", "g"], ["T+10ACGTACGTAC"] * 2]
    pool = ["A", "a", "C+1A", "c+1a", "C+2AG", "C+1T", "G-1N", "*", "T+11ACGTACGTACG", ">"]
    for _ in range(30):
        n = int(rng.integers(1, 40))
        cases.append([pool[int(k)] for k in rng.integers(0, len(pool), n)])
    out = []
    for toks in cases:
        bases, size = Events.ExtractInserts(_TokBam({7: toks} if toks is not None else {}), 7)
        out.append({"tokens": toks, "bases": bases, "size": size})
    bases, size = Events.ExtractInserts(_TokBam({}), 7)       # no column at all
    out.append({"tokens": None, "bases": bases, "size": size})
    return out


def golden_consensus(rng):
    """Sequences.BuildConsensus (+ ORFs.*) on planted-event count matrices."""
    cases = []
    for spec in ss.consensus_specs(rng):
        counts, orfs, region, mincov = spec["counts"], spec["orfs"], spec["region"], spec["mincov"]
        exp = {}
        for amb in (True, False):
            for ins in (True, False):
                exp["amb%d_ins%d" % (amb, ins)] = run_consensus(mincov, counts, orfs, amb, region, ins)
        idict = idict_from_counts(counts)
        li = Events.ListInserts(idict, mincov, _TokBam(region))
        cases.append({"name": spec["name"], "mincov": mincov,
                      "counts": [[int(v) for v in r] for r in counts],
                      "orfs": orfs, "region": {str(k): v for k, v in region.items()},
                      "inserts": li[1] and {str(k): v for k, v in li[1].items()},
                      "expected": exp})
    return cases


def golden_outputs(rng):
    """reads -> (oracle pileup emulator) -> unmodified BuildIndex -> WriteOutputs text."""
    out = []
    for spec in ss.read_specs(rng):
        reads = ss.reads_from_spec(spec)
        cols = orc.pileup_columns(reads)
        counts = orc.tally_matrix(reads, len(spec["ref"]))
        region = {}
        for i in range(len(counts)):
            if orc.insert_candidate(counts[i], spec["mincov"]):
                region[i + 1] = orc.region_tokens(reads, i + 1)
        path = spec["name"] + ".bam"
        _REG["bam"][path] = {"stageA": {c: t for c, t in cols.items()}, "region": region}
        _REG["fasta"]["ref.fa"] = ("refid", spec["ref"])
        df = indexing.BuildIndex(path, "ref.fa")
        idict = df.to_dict("index")
        got = np.array([[int(idict[p][c]) for c in orc.COLS] for p in sorted(idict)])
        assert sorted(idict) == list(range(1, len(got) + 1))
        res = {"name": spec["name"], "spec": spec, "counts": got.tolist(), "runs": {}}
        for amb in (True, False):
            key = "amb%d" % amb
            tmp = "/tmp/_tc_golden"
            os.makedirs(tmp, exist_ok=True)
            paths = {k: os.path.join(tmp, k) for k in ("fa", "vcf", "gff", "tsv")}
            for p in paths.values():
                if os.path.exists(p):
                    os.remove(p)

            class Hdr:
                raw_text = "##gff-version 3\n"
            sys.argv = ["TrueConsense", "ARGS"]
            try:
                Coverage.BuildCoverage(idict, paths["tsv"])
                Outputs.WriteOutputs(spec["mincov"], idict, gffdict_from_orfs(spec["orfs"], extra=True),
                                     path, amb, paths["vcf"], "SAMPLE", "ref.fa", paths["gff"], Hdr,
                                     paths["fa"])
                run = {k: open(p).read() for k, p in paths.items()}
                lines = run["vcf"].split("\n")
                lines[1] = "##fileDate=DATE"
                run["vcf"] = "\n".join(lines)
                # the corrected GFF as the command line writes it: rows as a GFF reader hands them over (nine columns, then
                # one key per attribute in file order) with seqid := sample name (TrueConsense.py:238-241)
                Outputs.WriteOutputs(spec["mincov"], idict, gffdict_cli(spec["orfs"]), path, amb, None, "SAMPLE", "ref.fa",
                                     paths["gff"], Hdr, paths["fa"])
                run["gff_cli"] = open(paths["gff"]).read()
            except (KeyError, ZeroDivisionError, IndexError) as e:
                run = {"raises": type(e).__name__}
            res["runs"][key] = run
        # --index-override through the reference's own functions (indexing.py:39-72, TrueConsense.py:232-235): three rows
        # of the index replaced, then the writers
        if len(got) >= 12 and "raises" not in res["runs"]["amb1"]:
            import pandas as pd
            rows = []
            for k, p1 in enumerate((2, len(got) // 2, len(got) - 1)):
                cov = 40 + 7 * k
                rows.append("%d,%d,%d,%d,%d,%d,%d,%d" % (p1, cov, cov - 9 - k, 4, 3, 2, k, (cov * 3) // 5 if k == 1 else 0))
            csv = ",coverage,A,T,C,G,X,I\n" + "\n".join(rows) + "\n"
            pz = "/tmp/_tc_golden_override_run.csv.gz"
            with gzip.open(pz, "wt") as fh:
                fh.write(csv)
            df2 = indexing.Override_index_positions(indexing.BuildIndex(path, "ref.fa"), indexing.read_override_index(pz))
            idict2 = df2.to_dict("index")
            try:
                Coverage.BuildCoverage(idict2, paths["tsv"])
                Outputs.WriteOutputs(spec["mincov"], idict2, gffdict_cli(spec["orfs"]), path, True, paths["vcf"], "SAMPLE", "ref.fa",
                                     paths["gff"], Hdr, paths["fa"])
                orun = {k: open(p).read() for k, p in paths.items()}
                lines = orun["vcf"].split("\n")
                lines[1] = "##fileDate=DATE"
                orun["vcf"] = "\n".join(lines)
            except (KeyError, ZeroDivisionError, IndexError) as e:
                orun = {"raises": type(e).__name__}
            orun["csv"] = csv
            res["override_run"] = orun
        out.append(res)
    return out


def golden_override():
    """indexing.read_override_index / Override_index_positions (indexing.py:39-72)."""
    import pandas as pd
    base = pd.DataFrame({c: [1, 2, 3, 4] for c in orc.COLS}, index=[1, 2, 3, 4])
    csv = ",coverage,A,T,C,G,X,I\n2,50,40,10,0,0,0,0\n4,9,9,0,0,0,0,7\n"
    p = "/tmp/_tc_golden_override.csv.gz"
    with gzip.open(p, "wt") as fh:
        fh.write(csv)
    over = indexing.read_override_index(p)
    merged = indexing.Override_index_positions(base.copy(), over).to_dict("index")
    return {"csv": csv, "base": base.values.tolist(),
            "merged": [[int(merged[i][c]) for c in orc.COLS] for i in (1, 2, 3, 4)]}


def main():
    rng = np.random.default_rng(20251121)
    sections = {
        "tokens": golden_tokens(rng),
        "rows": golden_rows(rng),
        "extract": golden_extract(rng),
        "consensus": golden_consensus(rng),
        "outputs": golden_outputs(rng),
        "override": golden_override(),
    }
    for name, data in sections.items():
        path = os.path.join(HERE, name + ".json")
        with open(path, "w") as fh:
            json.dump(data, fh, separators=(",", ":"))
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
