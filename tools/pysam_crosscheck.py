"""Cross-check of the ONE step of the path that no fixture in this repository pins: BAM bytes -> pileup tokens, which in the
reference happens inside pysam / htslib (indexing.py:100,139: `bamfile.pileup(stepper="nofilter", max_depth=10000000,
min_base_quality=0)` + `get_query_sequences(add_indels=True)`; Events.py:63-67: the default-argument region pileup behind
ExtractInserts).  pysam cannot be installed where this repository is built (no network), so oracle/tc_oracle.py RESTATES
htslib's pileup from its published algorithm and everything downstream is pinned by fixtures made
with the imported reference.  Anyone with pysam (the reference pins 0.23.3) can close the gap:

    python tools/pysam_crosscheck.py            # prints one line per case, exits 1 on the first difference
    python -m pytest tests/test_pysam_crosscheck.py        # the same as a test (skips itself where pysam is missing)

What it does: writes the BAM files the GPU tests already use (random CIGARs of every operation, indel carriers, a 9 500-deep
insert column, overlapping mates with insertions / deletions / ref-skips on the column), indexes them with pysam, and compares
  (a) every column's token list of the nofilter pileup with oracle.tc_oracle.pileup_columns, and
  (b) the default-argument region pileup's token list at the insert-candidate columns with oracle.tc_oracle.region_tokens
      (flag filter, orphans, base quality >= 13, max_depth = 8000, ignore_overlaps).
Token lists are compared in order (ExtractInserts' Counter.most_common tie-break is first-seen: Events.py:73-74)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    """-> [(name, reads dict, reference length, [1-based candidate columns])]"""
    from tests import fuzz_reads as fz
    from tests import synth_small as ss
    from trueconsense_amd import synthetic as sy
    out = []
    rng = np.random.default_rng(5)
    out.append(("fuzz: every CIGAR operation, odd SEQ content", fz.random_reads(rng, 3000, 2000), 2000, []))
    ref, orfs = sy.make_reference(L=6000, cds=[(10, 600)])
    sites = [(800, "D", 2, 0.5), (1500, "I", "AC", 0.7), (2500, "I", "ACGTACGTACGT", 0.6), (3000, "D", 1, 0.9)]
    indel = sy.make_reads(ref, 20000, seed=4, indel_sites=sites)
    indel["qual"] = np.random.default_rng(6).integers(0, 42, len(indel["qual"])).astype(np.uint8)
    out.append(("indel carriers, qualities 0..41", indel, len(ref), [800, 1500, 2500, 3000]))
    deep = []
    for k in range(9500):
        carrier = k < 4200
        deep.append({"pos": 100, "flag": 0, "cigar": "10M2I10M" if carrier else "20M",
                     "seq": "ACGTACGTAC" + ("GG" if carrier else "") + "ACGTACGTAC", "qual": 30})
    deep += [{"pos": 101 + j, "flag": 16, "cigar": "12M", "seq": "CGTACGTACACG", "qual": 35} for j in range(300)]
    deep.sort(key=lambda r: r["pos"])
    out.append(("a 9 500-deep insert column (max_depth = 8000)", ss.reads_from_spec({"reads": deep}), 600, [110]))
    rng = np.random.default_rng(8)
    pairs = []
    for k in range(400):
        start = 200 + int(rng.integers(0, 8))
        mate = start + int(rng.integers(0, 10))
        base1 = "ACGT"[int(rng.integers(0, 4))]
        base2 = base1 if rng.random() < 0.6 else "ACGT"[int(rng.integers(0, 4))]
        q1, q2 = int(rng.integers(5, 41)), int(rng.integers(5, 41))
        seq1 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30))
        seq2 = seq1[mate - start:] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, mate - start))
        i1, i2 = 214 - start, 214 - mate
        seq1 = seq1[:i1] + base1 + seq1[i1 + 1:]
        seq2 = seq2[:i2] + base2 + seq2[i2 + 1:]
        ins = rng.random() < 0.7
        cig1 = "%dM1I%dM" % (i1 + 1, 30 - i1 - 2) if ins else "30M"
        flags = (99, 147) if rng.random() < 0.9 else (65, 129)
        name = "pair%d" % k
        pairs.append({"pos": start, "flag": flags[0], "cigar": cig1, "seq": seq1, "qual": [q1] * 30, "name": name, "mtid": 0, "mpos": mate,
                      "tlen": mate + 30 - start})
        pairs.append({"pos": mate, "flag": flags[1], "cigar": "30M", "seq": seq2, "qual": [q2] * 30, "name": name, "mtid": 0, "mpos": start,
                      "tlen": -(mate + 30 - start)})
    pairs.sort(key=lambda r: r["pos"])
    out.append(("overlapping mates (ignore_overlaps)", ss.reads_from_spec({"reads": pairs}), 600, [214, 215, 216]))
    return out


def crosscheck(verbose=True):
    """-> number of differences (0: the restatement and pysam agree on every token list)."""
    import pysam
    from oracle import tc_oracle as orc
    from trueconsense_amd.io import bamwriter
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for name, reads, L, cand in cases():
            path = os.path.join(d, "x.bam")
            bamwriter.write_bam(path, reads, "ref", L, level=1)
            pysam.index(path)
            bam = pysam.AlignmentFile(path, "rb")
            want = orc.pileup_columns(reads)
            got = {p.pos: p.get_query_sequences(add_indels=True) for p in bam.pileup(stepper="nofilter", max_depth=10000000, min_base_quality=0)}
            diff = [c for c in sorted(set(want) | set(got)) if want.get(c) != got.get(c)]
            for pos1 in cand:
                toks = []
                for p in bam.pileup(bam.references[0], pos1 - 1, pos1, truncate=True):
                    toks = p.get_query_sequences(add_indels=True)
                if list(toks) != list(orc.region_tokens(reads, pos1)):
                    diff.append(("region", pos1))
            bam.close()
            if verbose:
                print("%-50s %6d columns, %d candidate columns: %s" % (name, len(want), len(cand), "agree" if not diff else "DIFFER at %r" % diff[:5]))
            bad += len(diff)
    return bad


if __name__ == "__main__":
    try:
        import pysam  # noqa: F401
    except ImportError:
        sys.exit("pysam is not installed here: run this where it is (pip install pysam==0.23.3)")
    sys.exit(1 if crosscheck() else 0)
