"""Command line — same flags, checks and exit codes as TrueConsense/TrueConsense.py:25-264.

    python -m trueconsense_amd.TrueConsense -i x.bam -ref r.fa -gff r.gff -cov 30 -name S -o out.fa
        [-vcf out.vcf] [-doc cov.tsv] [-ogff out.gff] [-t N] [-noambig] [--index-override f.csv.gz]

Additive flags (not in the reference): --device N (GPU ordinal), --stats FILE (JSON timings).
"""
from __future__ import annotations

import argparse
import json
import multiprocessing
import os
import pathlib
import sys
import time

from . import _state
from .Coverage import BuildCoverage
from .func import MyHelpFormatter, color
from .indexing import Gffindex, Override_index_positions, Readbam, build_counts, read_override_index
from .Outputs import WriteOutputs
from .version import __version__


def GetArgs(givenargs):
    """TrueConsense.py:25-209."""
    def isfile_or_exit(fname, code):
        if os.path.isfile(fname):
            return True
        print(f'"{fname}" is not a file. Exiting...')
        sys.exit(code)

    def checkbam(fname):
        isfile_or_exit(fname, -1)
        if pathlib.Path(fname).suffix != ".bam":
            parser.error(f"Input file {color.YELLOW}({fname}){color.END} doesn't seem to be a BAM-file.")
        return fname

    def checkfasta(fname):
        isfile_or_exit(fname, 1)
        if pathlib.Path(fname).suffix not in (".fasta", ".fa"):
            parser.error(f"Reference file {color.YELLOW}({fname}){color.END} doesn't seem to be a Fasta-file.")
        return fname

    def checkgff(fname):
        isfile_or_exit(fname, 1)
        if pathlib.Path(fname).suffix != ".gff":
            parser.error(f"Given file {color.YELLOW}({fname}){color.END} doesn't seem to be a GFF file.")
        return fname

    def check_index_override(fname):
        isfile_or_exit(fname, 1)
        ext = "".join(pathlib.Path(fname).suffixes)
        if ".csv" not in ext or ".gz" not in ext:
            parser.error(f"Given file {color.YELLOW}({fname}){color.END} doesn't seem to be a compressed csv file.")
        return fname

    parser = argparse.ArgumentParser(
        prog="TrueConsense", usage="%(prog)s [required options] [optional arguments]",
        description="TrueConsense: Creating biologically valid consensus sequences from reference-based alignments",
        formatter_class=MyHelpFormatter, add_help=False)
    standard_threads = min(multiprocessing.cpu_count(), 128)

    reqs = parser.add_argument_group("Required arguments")
    reqs.add_argument("--input", "-i", type=checkbam, metavar="File", help="Input file in BAM format", required=True)
    reqs.add_argument("--output", "-o", type=str, default=os.getcwd() + "consensus.fasta", metavar="File",
                      help="Output consensus fasta", required=True)
    reqs.add_argument("--reference", "-ref", type=checkfasta, metavar="File", help="Reference Fasta file",
                      required=True)
    reqs.add_argument("--features", "-gff", type=checkgff, metavar="File", help="File with genome features (GFF)",
                      required=True)
    reqs.add_argument("--coverage-level", "-cov", type=int, default=30, metavar="100",
                      help="The minimum coverage level of the consensus and variant calls", required=True)
    reqs.add_argument("--samplename", "-name", metavar="Text",
                      help="Name of the sample that is being processed, will be used to create the fasta header",
                      required=True)

    opts = parser.add_argument_group("Optional arguments")
    opts.add_argument("--variants", "-vcf", type=str, metavar="File", help="Output VCF file")
    opts.add_argument("--depth-of-coverage", "-doc", type=str, metavar="File",
                      help="Output TSV file listing the coverage per position")
    opts.add_argument("--output-gff", "-ogff", type=str, metavar="File", help="Ouput location a corrected GFF file")
    opts.add_argument("--threads", "-t", default=standard_threads, metavar="N", type=int,
                      help="Number of threads that can be used by TrueConsense")
    opts.add_argument("--noambiguity", "-noambig", action="store_true",
                      help="Turn off ambiguity nucleotides in the generated consensus sequence")
    opts.add_argument("--index-override", type=check_index_override, metavar="File",
                      help="Override the positional index of certain genome positions with 'known' information if "
                           "the given alignment is not sufficient for these positions\nMust be a compressed csv.\n"
                           "Please use with caution as this will overwrite the generated index at the given "
                           "positions!\n")
    opts.add_argument("--version", "-v", action="version", version=__version__,
                      help="Show the TrueConsense version and exit")
    opts.add_argument("--help", "-h", action="help", default=argparse.SUPPRESS,
                      help="Show this help message and exit")
    extra = parser.add_argument_group("MI355X arguments (additive)")
    extra.add_argument("--device", type=int, default=None, metavar="N", help="GPU ordinal (default: 0)")
    extra.add_argument("--stats", type=str, default=None, metavar="File", help="Write stage timings as JSON")
    return parser.parse_args(givenargs)


def main(args=None):
    """TrueConsense.py:212-264."""
    if not args:
        args = sys.argv[1:]
    if len(args) < 1:
        print("TrueConsense was called but no arguments were given, please try again.\n"
              "Use 'TrueConsense -h' to see the help document")
        sys.exit(1)
    a = GetArgs(args)
    if a.device is not None:
        os.environ["TCMI_DEVICE"] = str(a.device)
    t = {"start": time.perf_counter()}

    from .engine import BamFile
    bam = BamFile(a.input, threads=a.threads)
    t["bam_decode"] = time.perf_counter()
    counts = build_counts(bam, a.reference)
    IndexGff = Gffindex(a.features)
    t["tally"] = time.perf_counter()

    if a.index_override:
        import pandas as pd
        from ._ffi import COLS
        df = pd.DataFrame(counts.astype("int64"), columns=list(COLS), index=range(1, len(counts) + 1))
        df = Override_index_positions(df, read_override_index(a.index_override))
        counts = df.values
    indexDict = _state.IndexDict(counts)
    GffHeader = IndexGff.header
    GffDF = IndexGff.df
    GffDF["seqid"] = a.samplename
    GffDict = GffDF.to_dict("index")

    if a.depth_of_coverage is not None:
        BuildCoverage(indexDict, a.depth_of_coverage)

    IncludeAmbig = a.noambiguity is False
    WriteOutputs(a.coverage_level, indexDict, GffDict, bam, IncludeAmbig, a.variants, a.samplename, a.reference,
                 a.output_gff, GffHeader, a.output)
    t["outputs"] = time.perf_counter()
    if a.stats:
        keys = list(t)
        with open(a.stats, "w") as fh:
            json.dump({"seconds": {k: t[k] - t[keys[i - 1]] for i, k in enumerate(keys) if i},
                       "reads": bam.n_reads, "positions": len(counts), "bam_bytes": bam.file_bytes}, fh)


if __name__ == "__main__":
    main()
