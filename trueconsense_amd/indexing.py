"""Stage A host side — same surface as TrueConsense/indexing.py.

BuildIndex (indexing.py:75-154) keeps its signature and its return value (a DataFrame indexed
1..L with columns coverage,A,T,C,G,X,I) but the pileup and the per-token loop
(indexing.py:100-132) run as one HIP tally kernel over the decoded reads.
"""
from __future__ import annotations

from . import _state
from ._ffi import COLS
from . import _ffi
from .engine import BamFile, DeviceBam, LazyBam
from .io import fasta, gff


def Readbam(f):
    """indexing.py:6-19 — the BAM as the later stages see it (plays pysam.AlignmentFile's role downstream): decoded on the
    host only when somebody asks for reads (insert tokens)."""
    return f if isinstance(f, (BamFile, LazyBam)) or hasattr(f, "modal_token") else LazyBam(f)    # (modal_token: tokens resolved beforehand, e.g. gathered from several GPUs)


def Gffindex(file):
    """indexing.py:22-36 — object with `.df` (DataFrame) and `.header.raw_text`."""
    return gff.read_gff(file)


def read_override_index(f):
    """indexing.py:39-52."""
    import pandas as pd                                 # (where a DataFrame is the interface: imported when it is asked for)
    return pd.read_csv(f, sep=",", compression="gzip", index_col=0)


def Override_index_positions(index, override_data):
    """indexing.py:55-72."""
    index.loc[override_data.index, :] = override_data[:]
    return index


def build_counts(bamfile, ref, ctx=None):
    """BAM (+ reference FASTA, for its length) -> int32 [L,7] count matrix on the GPU path.  A path (or LazyBam) is decoded
    ON THE DEVICE (BGZF inflate, record chain, pack: csrc/bam_device.hip, pack_device.hip); files the device decoder
    declines, and BamFile objects, go through the host reader's flat arrays."""
    ref_length = fasta.first_length(ref) if isinstance(ref, str) else int(ref)
    ctx = ctx or _state.default_context()
    if not isinstance(bamfile, BamFile):
        path = bamfile.filename if isinstance(bamfile, LazyBam) else str(bamfile)
        d = DeviceBam(path)
        try:
            rs = ctx.upload_bamfile(d)
        except _ffi.TcmiError as e:
            if e.code != _ffi.E_UNSUPPORTED:
                raise
            rs = None
        finally:
            d.close()
        if rs is not None:
            try:
                build_counts.last_reads = int(rs.n_reads)
                return ctx.step(rs, max(ref_length, rs.max_end, 1), 0, True, want_counts=True)[3]
            finally:
                rs.free()
        bamfile = bamfile.get() if isinstance(bamfile, LazyBam) else BamFile(path)
    build_counts.last_reads = int(bamfile.n_reads)
    return ctx.tally(bamfile, ref_len=ref_length)


build_counts.last_reads = 0         # alignment records of the file the last call read (the command line's --stats)


def BuildIndex(bamfile, ref):
    """indexing.py:75-154."""
    counts = build_counts(bamfile, ref)
    import pandas as pd
    df = pd.DataFrame(counts.astype("int64"), columns=list(COLS), index=range(1, len(counts) + 1))
    df.index.name = None
    return df
