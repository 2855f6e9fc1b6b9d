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


_FIXED = [c for c in GFF3_COLUMNS if c != "attributes"]


def _gff_columns(row):
    """One GFF row dict -> ({fixed column: text} in canonical order, attributes text): the eight fixed columns, every other key
    folded into the attributes column (keys lower-cased; the parsed pairs of an `attributes` entry take its place in the row's
    key order)."""
    fixed, attrs = {}, {}
    for key, value in row.items():
        k = str(key).lower()
        if k == "attributes":
            for item in filter(None, str(value).split(";")):
                name, val = item.split("=")             # exactly one '=' per pair, or ValueError (as upstream)
                attrs[name] = val
        elif k in _FIXED:
            fixed[k] = str(value)
        else:
            attrs[k] = str(value)
    assert list(fixed) == _FIXED                        # upstream insists on the canonical column order, too
    return fixed, ";".join("%s=%s" % kv for kv in attrs.items())


def _gff_line(row):
    """One GFF row dict -> its text line."""
    fixed, attrs = _gff_columns(row)
    return "\t".join(list(fixed.values()) + [attrs]) + "\n"


def gff_row_columns(row):
    """-> [source, type, score, strand, phase, attributes] of a row: what the native batch writer (csrc/pipeline.cpp) puts around the
    sample's name and the corrected start / end."""
    fixed, attrs = _gff_columns(row)
    return [fixed["source"], fixed["type"], fixed["score"], fixed["strand"], fixed["phase"], attrs]


def vcf_header(today, argv, ref, refID):
    """The VCF's header lines (Outputs.py:115-127)."""
    return _VCF_HEAD.format(date=today, argv=" ".join(argv), ref=ref, contig=refID)


def WriteGFF(gffheader, gffdict, output_gff, name):
    """Outputs.py:13-71 — header text verbatim, then one line per row."""
    with open(output_gff, "w") as out:
        out.write(gffheader.raw_text)
        out.writelines(_gff_line(row) for row in gffdict.values())


_VCF_HEAD = ("##fileformat=VCFv4.3\n##fileDate={date}\n##source='TrueConsense {argv}'\n##reference='{ref}'\n##contig=<ID={contig}>\n"
             '##INFO=<ID=DP,Number=1,Type=Integer,Description="Read Depth">\n'
             '##INFO=<ID=INDEL,Number=0,Type=Flag,Description="Indicates that the variant is an INDEL.">\n'
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")


def vcf_text(today, argv, ref, refID, reflist, consensus_noinsert, iDict, mincov, hasinserts, insertpositions):
    """Outputs.py:115-180 as a single left-to-right scan over the reference: a run of '-' in the insert-free consensus yields
    one deletion record at its first index and is then jumped over; any other mismatch yields a SNP record; an index that
    equals an accepted insert's (1-based!) position yields an insertion record.  The quirks are upstream's:
    comparison against the raw reference characters, DP taken from the FOLLOWING position (index 0 and 1 both use position
    2), POS of indel records one less than their index + 1, IndexError when a deletion run reaches the end."""
    cons = consensus_noinsert.upper()
    ins = insertpositions if hasinserts is True else {}
    lines = [vcf_header(today, argv, ref, refID)]
    i, n = 0, len(reflist)
    while i < n:
        here = cons[i]
        step = 1
        if here == "-":
            j = i
            while cons[j] == "-":
                j += 1
            lines.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%s;INDEL\n"
                         % (refID, i, reflist[i - 1] + "".join(reflist[i:j]), cons[i - 1], GetCoverage(iDict, i + 1)))
            step = j - i
        elif here != reflist[i]:
            lines.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%s\n" % (refID, i + 1, reflist[i], here, GetCoverage(iDict, max(i, 1) + 1)))
        if i in ins:
            depth = GetCoverage(iDict, i + 1)
            if depth > mincov:
                for bases in ins[i].values():
                    lines.append("%s\t%d\t.\t%s\t%s\t.\tPASS\tDP=%s;INDEL\n" % (refID, i, reflist[i], here + str(bases), depth))
        i += step
    return "".join(lines)


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
