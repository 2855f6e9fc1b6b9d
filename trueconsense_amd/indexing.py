"""Stage A host side — same surface as TrueConsense/indexing.py.

BuildIndex (indexing.py:75-154) keeps its signature and its return value (a DataFrame indexed
1..L with columns coverage,A,T,C,G,X,I) but the pileup and the per-token loop
(indexing.py:100-132) run as one HIP tally kernel over the decoded reads.
"""
from __future__ import annotations

import pandas as pd

from . import _state
from ._ffi import COLS
from .engine import BamFile
from .io import fasta, gff


def Readbam(f):
    """indexing.py:6-19 — the decoded BAM (plays pysam.AlignmentFile's role downstream)."""
    return f if isinstance(f, BamFile) else BamFile(f)


def Gffindex(file):
    """indexing.py:22-36 — object with `.df` (DataFrame) and `.header.raw_text`."""
    return gff.read_gff(file)


def read_override_index(f):
    """indexing.py:39-52."""
    return pd.read_csv(f, sep=",", compression="gzip", index_col=0)


def Override_index_positions(index, override_data):
    """indexing.py:55-72."""
    index.loc[override_data.index, :] = override_data[:]
    return index


def build_counts(bamfile, ref, ctx=None):
    """BAM (+ reference FASTA, for its length) -> int32 [L,7] count matrix on the GPU path."""
    bam = Readbam(bamfile)
    ref_length = fasta.first_length(ref) if isinstance(ref, str) else int(ref)
    ctx = ctx or _state.default_context()
    return ctx.tally(bam, ref_len=ref_length)


def BuildIndex(bamfile, ref):
    """indexing.py:75-154."""
    counts = build_counts(bamfile, ref)
    df = pd.DataFrame(counts.astype("int64"), columns=list(COLS), index=range(1, len(counts) + 1))
    df.index.name = None
    return df
