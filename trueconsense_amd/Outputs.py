"""Writers — same surface as TrueConsense/Outputs.py (WriteGFF :13-71, WriteOutputs :74-183).

The text formats and their quirks follow the reference exactly: the VCF
compares the raw reference characters with the upper-cased insert-free consensus, emits every
low-coverage `N` as a SNP, takes DP from the *following* position, and matches insert records
by 0-based index == 1-based insert position.
"""
from __future__ import annotations

import sys
from datetime import date

from . import _state
from .Coverage import GetCoverage
from .Events import inserts_from_flags
from .indexing import Readbam
from .io import fasta
from .io.gff import GFF3_COLUMNS
from .Sequences import consensus_from_records


def WriteGFF(gffheader, gffdict, output_gff, name):
    """Outputs.py:13-71 — nine tab-separated columns; every other key is lower-cased and folded
    into `attributes` after the parsed attribute pairs."""
    cols_without_attr = [c for c in GFF3_COLUMNS if c != "attributes"]

    def fold(row):
        clean, extra = {}, {}
        for k, v in row.items():
            lk = str(k).lower()
            if lk not in cols_without_attr:
                extra[lk] = str(v)
            else:
                clean[lk] = str(v)
        attrs = {}
        for k, v in extra.items():
            if k == "attributes":
                for item in v.split(";"):
                    if item == "":
                        continue
                    key, value = item.split("=")
                    attrs[key] = value
            else:
                attrs[k] = v
        clean["attributes"] = ";".join("%s=%s" % kv for kv in attrs.items())
        assert list(clean.keys()) == GFF3_COLUMNS
        return clean

    with open(output_gff, "w") as out:
        out.write(gffheader.raw_text)
        for _, row in gffdict.items():
            out.write("\t".join(str(v) for v in fold(row).values()) + "\n")


def vcf_text(today, argv, ref, refID, reflist, consensus_noinsert, iDict, mincov, hasinserts, insertpositions):
    """Outputs.py:115-180."""
    seqlist = list(consensus_noinsert.upper())
    out = [f"""##fileformat=VCFv4.3
##fileDate={today}
##source='TrueConsense {' '.join(argv)}'
##reference='{ref}'
##contig=<ID={refID}>
##INFO=<ID=DP,Number=1,Type=Integer,Description="Read Depth">
##INFO=<ID=INDEL,Number=0,Type=Flag,Description="Indicates that the variant is an INDEL.">
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO
"""]
    delskips = set()
    for i in range(len(reflist)):
        if i in delskips:
            continue
        if reflist[i] != seqlist[i]:
            if seqlist[i] == "-":
                b = i
                gone = []
                while seqlist[b] == "-":            # IndexError past the end, like the reference
                    gone.append(reflist[b])
                    delskips.add(b)
                    b += 1
                currentcov = GetCoverage(iDict, i + 1)
                out.append(f"{refID}\t{i}\t.\t{reflist[i - 1] + ''.join(gone)}\t{seqlist[i - 1]}\t.\tPASS\t"
                           f"DP={currentcov};INDEL\n")
            else:
                p = 1 if i < 2 else i
                currentcov = GetCoverage(iDict, p + 1)
                out.append(f"{refID}\t{i + 1}\t.\t{reflist[i]}\t{seqlist[i]}\t.\tPASS\tDP={currentcov}\n")
        if hasinserts is True and i in insertpositions:
            currentcov = GetCoverage(iDict, i + 1)
            if currentcov > mincov:
                for y in insertpositions.get(i):
                    to_insert = str(insertpositions.get(i).get(y))
                    out.append(f"{refID}\t{i}\t.\t{reflist[i]}\t{seqlist[i] + to_insert}\t.\tPASS\t"
                               f"DP={currentcov};INDEL\n")
    return "".join(out)


def WriteOutputs(mincov, iDict, uGffDict, inputbam, IncludeAmbig, output_vcf, name, ref, output_gff, gffheader,
                 output_consensus):
    """Outputs.py:74-183.  The call records are computed once on the GPU and shared by the two
    consensus walks (with / without inserts) and the VCF's insert list."""
    today = date.today().strftime("%Y%m%d")
    bam = Readbam(inputbam)
    counts = _state.counts_of(iDict)
    plain, alt, flags = _state.default_context().call(counts, mincov, IncludeAmbig)
    hasinserts, insertpositions = inserts_from_flags(flags, bam)
    consensus, newgff = consensus_from_records(plain, alt, flags, uGffDict, insertpositions, True)
    consensus_noinsert = consensus_from_records(plain, alt, flags, uGffDict, insertpositions, False)[0]

    if output_gff is not None:
        WriteGFF(gffheader, newgff, output_gff, name)

    if output_vcf is not None:
        refID, refseq = fasta.read_first_record(ref)
        text = vcf_text(today, sys.argv[1:], ref, refID, list(refseq), consensus_noinsert, iDict, mincov,
                        hasinserts, insertpositions)
        with open(output_vcf, "w") as out:
            out.write(text)

    with open(output_consensus, "w") as out:
        out.write(f">{name} mincov={mincov}\n{consensus}\n")
