"""The pileup semantics that live in pysam / htslib (indexing.py:100,139; Events.py:63-67) are RESTATED in oracle/tc_oracle.py and
not pinned by any fixture here — pysam cannot be installed in the build container.  Where pysam is importable this test closes
that gap (tools/pysam_crosscheck.py says how); elsewhere it skips itself."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_token_lists_equal_pysams():
    pytest.importorskip("pysam")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pysam_crosscheck
    assert pysam_crosscheck.crosscheck(verbose=True) == 0


def test_the_crosscheck_cases_build_without_pysam():
    """(the inputs of the cross-check are well-formed here: they go through the BAM writer, the host reader and the oracle)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pysam_crosscheck
    from oracle import tc_oracle as orc
    n = 0
    for name, reads, L, cand in pysam_crosscheck.cases():
        assert int(reads["n_reads"]) > 0 and L > 0
        for pos1 in cand:
            n += len(orc.region_tokens(reads, pos1))
    assert n > 8000
