"""Command line — same flags, checks and exit codes as TrueConsense/TrueConsense.py:25-264.

    python -m trueconsense_amd.TrueConsense -i x.bam -ref r.fa -gff r.gff -cov 30 -name S -o out.fa
        [-vcf out.vcf] [-doc cov.tsv] [-ogff out.gff] [-t N] [-noambig] [--index-override f.csv.gz]

Additive flags (not in the reference): --device N (GPU ordinal), --stats FILE (JSON timings), and

    python -m trueconsense_amd.TrueConsense --batch MANIFEST -ref r.fa -gff r.gff -cov 30 [-noambig] [-t N]

for many samples against one reference in one process: MANIFEST holds one sample per line, tab-separated:
input BAM, sample name, consensus FASTA and — optional, "-" or empty for "not wanted" — VCF, corrected GFF, coverage TSV
(the -i / -name / -o / -vcf / -ogff / -doc of the single-sample form).  The samples go through the native file runner
(csrc/pipeline.cpp): reading, GPU work and the host walk of consecutive samples overlap, all four outputs are written natively.
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
from .indexing import Gffindex, Override_index_positions, build_counts, read_override_index
from .Outputs import WriteOutputs
from .version import __version__


def _file_arg(parser, suffixes, what, missing_exit, multi_suffix=False):
    """argparse `type=` callable: the path must exist (else print + exit with the reference's code,
    TrueConsense.py:26-73) and carry one of `suffixes` (else parser.error, exit code 2)."""
    def check(fname):
        if not os.path.isfile(fname):
            print(f'"{fname}" is not a file. Exiting...')
            sys.exit(missing_exit)
        p = pathlib.Path(fname)
        ext = "".join(p.suffixes) if multi_suffix else p.suffix
        ok = all(sfx in ext for sfx in suffixes) if multi_suffix else ext in suffixes
        if not ok:
            parser.error(f"{what[0]} {color.YELLOW}({fname}){color.END} doesn't seem to be {what[1]}.")
        return fname
    return check


def GetArgs(givenargs):
    """Same flags, defaults, required-ness and exit codes as TrueConsense.py:25-209; table-driven."""
    parser = argparse.ArgumentParser(
        prog="TrueConsense", usage="%(prog)s [required options] [optional arguments]",
        description="TrueConsense: Creating biologically valid consensus sequences from reference-based alignments",
        formatter_class=MyHelpFormatter, add_help=False)
    bam_t = _file_arg(parser, (".bam",), ("Input file", "a BAM-file"), -1)
    fasta_t = _file_arg(parser, (".fasta", ".fa"), ("Reference file", "a Fasta-file"), 1)
    gff_t = _file_arg(parser, (".gff",), ("Given file", "a GFF file"), 1)
    csvgz_t = _file_arg(parser, (".csv", ".gz"), ("Given file", "a compressed csv file"), 1, multi_suffix=True)
    threads = min(multiprocessing.cpu_count(), 128)
    #          flags                          keyword arguments
    required = [
        (("--input", "-i"), dict(type=bam_t, metavar="File", help="alignment to call the consensus from (BAM)")),
        (("--output", "-o"), dict(type=str, default=os.getcwd() + "consensus.fasta", metavar="File",
                                  help="where the consensus FASTA goes")),
        (("--reference", "-ref"), dict(type=fasta_t, metavar="File", help="reference sequence (FASTA)")),
        (("--features", "-gff"), dict(type=gff_t, metavar="File", help="genome features of the reference (GFF)")),
        (("--coverage-level", "-cov"), dict(type=int, default=30, metavar="100",
                                            help="minimum coverage for a position to be called")),
        (("--samplename", "-name"), dict(metavar="Text", help="sample name, used in the FASTA header")),
    ]
    optional = [
        (("--variants", "-vcf"), dict(type=str, metavar="File", help="also write a VCF")),
        (("--depth-of-coverage", "-doc"), dict(type=str, metavar="File", help="also write position<TAB>coverage (TSV)")),
        (("--output-gff", "-ogff"), dict(type=str, metavar="File", help="also write the corrected GFF")),
        (("--threads", "-t"), dict(default=threads, metavar="N", type=int, help="host threads (BAM decoding, packing)")),
        (("--noambiguity", "-noambig"), dict(action="store_true", help="no IUPAC ambiguity codes in the consensus")),
        (("--index-override",), dict(type=csvgz_t, metavar="File",
                                     help="gzipped CSV (position,coverage,A,T,C,G,X,I) whose rows replace the tallied\n"
                                          "counts at those positions; use with caution")),
        (("--version", "-v"), dict(action="version", version=__version__, help="print the version and exit")),
        (("--help", "-h"), dict(action="help", default=argparse.SUPPRESS, help="print this help and exit")),
    ]
    additive = [
        (("--device",), dict(type=int, default=None, metavar="N", help="GPU ordinal (default: 0)")),
        (("--stats",), dict(type=str, default=None, metavar="File", help="write stage timings as JSON")),
    ]
    additive.append((("--batch",), dict(type=str, default=None, metavar="File",
                                        help="many samples: a tab-separated manifest (BAM, name, FASTA[, VCF, GFF, TSV] per line)\n"
                                             "instead of -i / -name / -o / -vcf / -ogff / -doc")))
    additive.append((("--gpus",), dict(type=int, default=1, metavar="N",
                                       help="GPUs of this node to use: with --batch the manifest's samples are dealt to N processes,\n"
                                            "one per GPU (independent files, no exchange); with -i ONE BAM file is shared — every GPU\n"
                                            "takes a range of its BGZF blocks, one reduce of the count matrix")))
    # (is this the --batch form?  asked of a small parser of its own: "--batch=FILE" and argparse's abbreviations count too)
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--batch", default=None)
    batch = pre.parse_known_args(givenargs)[0].batch is not None
    for title, rows, req in (("Required arguments", required, True), ("Optional arguments", optional, False),
                             ("MI355X arguments (additive)", additive, False)):
        group = parser.add_argument_group(title)
        for flags, kw in rows:
            if req:
                kw["required"] = not (batch and flags[0] in ("--input", "--output", "--samplename"))
            group.add_argument(*flags, **kw)
    return parser.parse_args(givenargs)


def _spawn(cmds, envs):
    """Start the children BEFORE this process touches a GPU (never re-exec a process that did), wait for all, -> worst exit code.
    The children are polled together: once one of them has failed the others get a short grace (they may be waiting for it in a
    collective) and are then terminated, so nobody sits out a collective's time-out."""
    import subprocess
    procs = [subprocess.Popen(c, env=e) for c, e in zip(cmds, envs)]
    rcs = [None] * len(procs)
    failed_at = None
    while any(r is None for r in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed_at is None:
                    failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > float(os.environ.get("TCMI_SPAWN_GRACE", "20")):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.02)
    return max((abs(r) for r in rcs), default=0)


def _child_argv(a, single):
    """The command line of a --gpus child, built from the PARSED namespace (argparse takes abbreviations, so the spelling the user typed
    cannot be filtered by name): the reference's flags as they were understood, never --gpus / --batch / --device / --stats — the
    caller adds its own — plus an explicit `--gpus 1`, so that a child can never deal itself out again."""
    out = ["-ref", a.reference, "-gff", a.features, "-cov", str(a.coverage_level), "-t", str(a.threads)]
    if a.noambiguity:
        out.append("-noambig")
    if single:
        out += ["-i", a.input, "-o", a.output, "-name", a.samplename]
        for flag, v in (("-vcf", a.variants), ("-doc", a.depth_of_coverage), ("-ogff", a.output_gff)):
            if v is not None:
                out += [flag, v]
    return out + ["--gpus", "1"]


def run_gpus(a):
    """--gpus N: N processes, one per GPU.  --batch: the manifest's samples dealt round-robin (BASELINE configs[3]: independent files,
    no collective); -i: ONE BAM file shared by the GPUs (configs[4]: trueconsense_amd.split_main)."""
    import socket
    import tempfile
    n = int(a.gpus)
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                    PYTHONPATH=os.pathsep.join([pkg_parent] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    one_gpu = os.environ.get("TCMI_SPLIT_ONE_GPU") == "1"              # (rehearsal on a one-GPU box: every child on GPU 0)
    if a.batch:
        rows = [ln for ln in open(a.batch).read().split("\n") if ln.strip() and not ln.startswith("#")]
        with tempfile.TemporaryDirectory(prefix="tcmi_gpus_") as tmp:
            cmds, envs = [], []
            for k in range(n):
                mine = rows[k::n]
                if not mine:
                    continue
                shard = os.path.join(tmp, "shard%d.tsv" % k)
                with open(shard, "w") as fh:
                    fh.write("\n".join(mine) + "\n")
                cmd = [sys.executable, "-m", "trueconsense_amd.TrueConsense", "--batch", shard, "--device", str(0 if one_gpu else k)] + _child_argv(a, False)
                if a.stats:
                    cmd += ["--stats", "%s.gpu%d" % (a.stats, k)]
                cmds.append(cmd)
                envs.append(base_env)
            return _spawn(cmds, envs)
    if a.index_override:
        print("--index-override goes with one GPU. Exiting...")
        return 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmds, envs = [], []
    for k in range(n):
        cmds.append([sys.executable, "-m", "trueconsense_amd.split_main"] + _child_argv(a, True))
        envs.append(dict(base_env, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    return _spawn(cmds, envs)


def run_batch(a):
    """--batch: the manifest's samples through the native file runner, four output files each."""
    from datetime import date
    from .engine import FileRunner
    from .io import fasta
    from .Outputs import gff_row_columns, vcf_header
    if a.index_override:
        print("--index-override goes with a single sample (-i), not with --batch. Exiting...")
        sys.exit(1)
    rows = []
    with open(a.batch) as fh:
        for ln, line in enumerate(fh, 1):
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            f = line.split("\t")
            if len(f) < 3:
                print(f'{a.batch}:{ln}: need at least "BAM<TAB>name<TAB>FASTA". Exiting...')
                sys.exit(1)
            f += [""] * (6 - len(f))
            if not os.path.isfile(f[0]):
                print(f'"{f[0]}" is not a file. Exiting...')
                sys.exit(-1)
            rows.append([f[0], f[1], f[2]] + [None if x in ("", "-") else x for x in f[3:6]])
    IndexGff = Gffindex(a.features)
    gffrows = list(IndexGff.index_dict(seqid="S").values())       # (the runner puts each sample's name there: TrueConsense.py:240)
    refID, refseq = fasta.read_first_record(a.reference)
    t0 = time.perf_counter()
    cores = max(1, min(int(a.threads), os.cpu_count() or 1))
    runner = FileRunner(int(os.environ.get("TCMI_DEVICE", "0")), gffrows, a.coverage_level, a.noambiguity is False,
                        decoders=min(4, max(1, cores // 4)), decode_threads=max(1, cores // 2), walkers=min(4, max(1, cores // 4)),
                        gpu_streams=(8 if len(rows) > 16 else 3) if len(rows) > 2 else 1)
    runner.set_outputs(refID, refseq, vcf_header(date.today().strftime("%Y%m%d"), sys.argv[1:], a.reference, refID), IndexGff.header.raw_text,
                       [gff_row_columns(r) for r in gffrows])
    try:
        runner.run_files([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], [r[3] for r in rows], [r[4] for r in rows],
                         [r[5] for r in rows], ref_len=len(refseq))
    except Exception:                                                # every failed sample is named; the first one's error (what the reference would raise) goes on up: non-zero exit
        failed = [(rows[i][1], int(c)) for i, c in enumerate(getattr(runner, "last_status", [])) if int(c) != 0]
        for nm, code in failed:
            print(f'sample "{nm}" failed (libtcmi error {code})', file=sys.stderr)
        if failed:
            print(f"{len(failed)} of {len(rows)} samples failed", file=sys.stderr)
        raise
    finally:
        if a.stats:
            with open(a.stats, "w") as fh:
                json.dump({"seconds": {"batch": time.perf_counter() - t0}, "samples": len(rows), "stage_busy_seconds": runner.seconds,
                           "decoded_on": runner.decoded_on, "status": [int(x) for x in getattr(runner, "last_status", [])]}, fh)
        runner.close()


def main(args=None):
    """TrueConsense.py:212-264."""
    if not args:
        args = sys.argv[1:]
    if len(args) < 1:
        print("TrueConsense was called but no arguments were given, please try again.\n"
              "Use 'TrueConsense -h' to see the help document")
        sys.exit(1)
    a = GetArgs(args)
    if a.gpus and a.gpus > 1:
        rc = run_gpus(a)
        if rc:
            sys.exit(rc)
        return
    if a.device is not None:
        os.environ["TCMI_DEVICE"] = str(a.device)
    if a.batch:
        return run_batch(a)
    t = {"start": time.perf_counter()}

    from .engine import LazyBam
    bam = LazyBam(a.input, threads=a.threads)       # reads reach the host only if an insert candidate needs its tokens
    t["bam_open"] = time.perf_counter()
    counts = build_counts(bam, a.reference)         # decoded, packed and tallied on the device
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
    GffDict = IndexGff.index_dict(seqid=a.samplename)           # (TrueConsense.py:238-241: df["seqid"] = samplename; df.to_dict("index") — without importing pandas)

    if a.depth_of_coverage is not None:
        BuildCoverage(indexDict, a.depth_of_coverage)

    IncludeAmbig = a.noambiguity is False
    WriteOutputs(a.coverage_level, indexDict, GffDict, bam, IncludeAmbig, a.variants, a.samplename, a.reference,
                 a.output_gff, GffHeader, a.output)
    t["outputs"] = time.perf_counter()
    if a.stats:
        keys = list(t)
        with open(a.stats, "w") as fh:
            secs = {k: t[k] - t[keys[i - 1]] for i, k in enumerate(keys) if i}
            secs["bam_decode"] = secs["bam_open"]       # (round 1's name of the same span: the file is opened, decoded with the tally)
            json.dump({"seconds": secs, "positions": len(counts), "reads": build_counts.last_reads,
                       "bam_bytes": os.path.getsize(a.input)}, fh)


if __name__ == "__main__":
    main()
