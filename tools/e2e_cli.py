#!/usr/bin/env python3
"""End to end on the GPU box: write a synthetic 1M-read BAM + FASTA + GFF, run the command line
(BAM decode -> H2D -> tally/call kernels -> host walks -> FASTA/VCF/GFF/TSV) and print stage timings,
including the PCIe-inclusive rate that bench.py (reads resident in HBM) leaves out."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                    # noqa: E402
from trueconsense_amd import synthetic as sy                           # noqa: E402
from trueconsense_amd.io import bamwriter                              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    indels = len(sys.argv) > 2 and sys.argv[2] == "indels"
    d = tempfile.mkdtemp(prefix="tcmi_e2e_")
    ref, orfs = sy.make_reference()
    t = time.perf_counter()
    reads = sy.make_reads(ref, n, seed=5, indel_sites=sy.default_indel_sites(orfs) if indels else None)
    t_gen = time.perf_counter() - t
    t = time.perf_counter()
    if indels:
        bamwriter.write_bam(os.path.join(d, "in.bam"), reads, "MN908947.3", len(ref))
    else:
        bamwriter.write_bam_fast(os.path.join(d, "in.bam"), reads["pos"], reads["flag"],
                                 reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref))
    t_write = time.perf_counter() - t
    with open(os.path.join(d, "ref.fa"), "w") as fh:
        fh.write(">MN908947.3 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
    head, body = sy.gff_text(orfs)
    with open(os.path.join(d, "f.gff"), "w") as fh:
        fh.write(head + body)
    from trueconsense_amd import TrueConsense as cli
    from trueconsense_amd.engine import BamFile, Context
    argv = ["-i", os.path.join(d, "in.bam"), "-ref", os.path.join(d, "ref.fa"), "-gff", os.path.join(d, "f.gff"),
            "-cov", "30", "-name", "S", "-o", os.path.join(d, "out.fa"), "-vcf", os.path.join(d, "out.vcf"),
            "-ogff", os.path.join(d, "out.gff"), "-doc", os.path.join(d, "out.tsv"), "--stats", os.path.join(d, "stats.json")]
    runs = []
    for _ in range(3):
        t = time.perf_counter()
        cli.main(argv)
        runs.append((time.perf_counter() - t, json.load(open(os.path.join(d, "stats.json")))))
    # H2D-inclusive hot path: decoded reads on the host -> upload -> step -> records
    bam = BamFile(os.path.join(d, "in.bam"))
    ctx = Context(0)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        rs = ctx.upload(bam)
        t1 = time.perf_counter()
        ctx.step(rs, len(ref), 30, True, want_counts=False)
        ts.append((t1 - t, time.perf_counter() - t1))
        rs.free()
    # sanity of the file path (parity itself is the tests' job): every base is one pileup token, deletions add theirs
    got = np.loadtxt(os.path.join(d, "out.tsv"), dtype=np.int64)[:, 1]
    cg = np.asarray(reads["cigar"], np.int64)
    tokens = int(((cg >> 4) * np.isin(cg & 15, (0, 2, 3, 7, 8))).sum())
    out = {"reads": n, "bam_bytes": os.path.getsize(os.path.join(d, "in.bam")), "generate_s": t_gen, "write_bam_s": t_write,
           "cli_wall_s": [r[0] for r in runs], "cli_stages_s": runs[-1][1]["seconds"],
           "upload_pack_h2d_s": [a for a, _ in ts], "step_s": [b for _, b in ts],
           "positions_per_s_cli": len(ref) / min(r[0] for r in runs),
           "positions_per_s_host_resident_reads": len(ref) / min(a + b for a, b in ts),
           "coverage_sum_equals_pileup_tokens": bool(int(got.sum()) == tokens),
           "fasta_len": len(open(os.path.join(d, "out.fa")).read().split("\n")[1]), "cpus": os.cpu_count()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
