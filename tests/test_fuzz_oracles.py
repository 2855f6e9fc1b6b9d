"""Differential fuzz on the CPU: the column-major Python emulator (tc_oracle.pileup_columns fed to
the reference's own token rule) against the independent read-major C restatement, on random reads
with arbitrary CIGARs.  This is what stands in for pysam at the one step that cannot be pinned."""
import numpy as np

from oracle import c_oracle
from oracle import tc_oracle as orc
from tests import fuzz_reads as fz


def test_python_emulator_equals_c_oracle_on_random_cigars():
    rng = np.random.default_rng(424242)
    for rep in range(12):
        L = int(rng.integers(200, 900))
        reads = fz.random_reads(rng, 250, L, long_reads=(rep % 4 == 3))
        Lx = c_oracle.extent(reads, L)
        want = orc.tally_matrix(reads, L)
        assert len(want) == Lx
        got = c_oracle.tally(reads, Lx)
        assert np.array_equal(got, want), rep
        assert got[:, 5].sum() > 0 and got[:, 6].sum() > 0          # deletions and insertions occurred
