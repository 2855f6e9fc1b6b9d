"""First-record FASTA access — the two things the reference asks of a FASTA file:
pysam.FastaFile(ref).lengths[0] (indexing.py:97-98) and the first Bio.SeqIO record's
id / sequence (Outputs.py:108-113)."""
from __future__ import annotations


def read_first_record(path):
    """-> (id, sequence).  id = header text up to the first whitespace (Bio.SeqIO semantics)."""
    rid, chunks = None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if rid is not None:
                    break
                parts = line[1:].split(None, 1)
                rid = parts[0] if parts else ""
            elif rid is not None:
                chunks.append(line.strip())
    if rid is None:
        raise ValueError("%s holds no FASTA record" % path)
    return rid, "".join(chunks)


def first_length(path):
    return len(read_first_record(path)[1])
