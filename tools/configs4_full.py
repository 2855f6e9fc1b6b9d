#!/usr/bin/env python3
"""BASELINE configs[4] at its stated size on the one GPU there is: ONE BAM file of `--reads` x `--world` reads (default 6.25 M x 8 =
50 M reads, 13.6 GB inflated), its `--world` block ranges decoded, packed and tallied ONE AFTER THE OTHER through tcmi_split_step
(distributed.split_ranks_in_turn: every rank's step exactly as a rank of the 8-GPU job runs it — range + the block behind it, range
table, pairwise joins on the root, call on the root — with the reduce played by an accumulator on the device), then vote + walk.
Checked against the oracle (outside any clock): counts = the scalar C tally of every read written, FASTA = the oracle chain.
The projected step of an 8-GPU node = max(rank seconds) + the reduce of 0.84 MB.

    python3 tools/configs4_full.py [--reads 6250000] [--world 8] [--repeat 2] [--json OUT]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6_250_000, help="reads per rank's tile")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=2, help="passes over the ranks (the first one allocates)")
    ap.add_argument("--mincov", type=int, default=30)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--json", default=None)
    ap.add_argument("--split-sub", default=None, help="sub-ranges per rank in tcmi_split_step (default: the library's choice); a list \"1,3,0\": the passes once per value, turn about")
    a = ap.parse_args()
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    ref, orfs = sy.make_reference()
    L = len(ref)
    tile = L - 150 + 1
    tmp = tempfile.mkdtemp(prefix="tcmi_c4_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(tmp, "configs4.bam")
    total = a.reads * a.world
    want = np.zeros((L, 7), np.int64)
    t0 = time.time()
    piece = 1_000_000                                                # written a million reads at a time, in coordinate order
    n_pieces = (total + piece - 1) // piece
    done = 0
    for k in range(n_pieces):
        n = min(piece, total - done)
        reads = sy.make_reads(ref, n, seed=9000 + k, start_range=(tile * k // n_pieces, tile * (k + 1) // n_pieces))
        bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=a.level,
                                 part=(k == 0, k == n_pieces - 1), first_id=done)
        want += c_oracle.tally(reads, L)
        done += n
        if k % 10 == 9:
            print("written %d M reads, %.0f s" % (done // 1_000_000, time.time() - t0), flush=True)
    print("file: %d reads, %.1f MB, written in %.0f s" % (total, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
    rows = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs]
    out = {"reads": total, "world": a.world, "passes": []}
    text = counts = None
    subs = [None] if a.split_sub is None else [int(x) for x in str(a.split_sub).split(",")]
    for rep, sub_ranges in [(r_, s_) for r_ in range(a.repeat) for s_ in subs]:
        tm = {}
        t1 = time.time()
        text, counts, toks = td.split_ranks_in_turn(path, L, rows, a.mincov, a.world, True, "S", return_parts=True, timings=tm, split_sub=sub_ranges)
        ms = [1e3 * s for s in tm["rank_seconds"]]
        print("pass %d (split_sub %s, taken %d): per-rank tcmi_split_step ms %s | max %.2f sum %.2f | whole call %.2f s | blocks per rank %s | batches of blocks: %s, one-sync path: %d of %d"
              % (rep, sub_ranges, tm["split_sub_taken"], " ".join("%.2f" % m for m in ms), max(ms), sum(ms), time.time() - t1, tm["blocks_per_rank"], tm["decode_batched"] > 0, tm["one_sync_taken"], a.world), flush=True)
        out["passes"].append({"split_sub": sub_ranges, "rank_ms": ms, "max_ms": max(ms), "sum_ms": sum(ms), "blocks_per_rank": tm["blocks_per_rank"], "decode_batched": tm["decode_batched"],
                              "one_sync_taken": tm["one_sync_taken"], "split_sub_taken": tm["split_sub_taken"]})
        out.update(file_bytes=tm["file_bytes"], inflated_bytes=tm["inflated_bytes"], n_blocks=tm["n_blocks"])
    print("file %.1f MB, %.2f GiB inflated, %d BGZF blocks" % (out["file_bytes"] / 1e6, out["inflated_bytes"] / 2 ** 30, out["n_blocks"]))
    eq = bool(np.array_equal(counts.astype(np.int64), want))
    print("counts equal the oracle's:", eq, "| coverage sum", int(counts[:, 0].sum()), "expected", 150 * total, "| max depth", int(counts[:, 0].max()), flush=True)
    t2 = time.time()
    has, ins = orc.list_inserts(want, a.mincov, lambda pos1: [])
    cons, _ = orc.build_consensus(a.mincov, want, [dict(o) for o in orfs], True, ins if has else None, True)
    want_text = orc.fasta_text("S", a.mincov, cons)
    fe = bool(text == want_text)
    print("FASTA equals the oracle chain's:", fe, "(oracle chain %.1f s)" % (time.time() - t2))
    best = min(out["passes"][1:] or out["passes"], key=lambda p: p["max_ms"])
    ld = (L + 255) // 256 * 256
    print("projected step of a %d-GPU node: max(rank) %.2f ms + the reduce of %d bytes (one rank's GPU does %.2f ms of the file's %.2f ms)"
          % (a.world, best["max_ms"], 4 * (7 * ld + 6 * a.world + 1), best["max_ms"], best["sum_ms"]))
    out.update(counts_bit_exact=eq, fasta_bit_exact=fe, projected_step_ms_without_the_reduce=best["max_ms"], coverage_max=int(counts[:, 0].max()))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
    os.remove(path)
    os.rmdir(tmp)
    sys.exit(0 if eq and fe else 1)


if __name__ == "__main__":
    main()
