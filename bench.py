#!/usr/bin/env python3
"""bench.py — throughput of the pileup-tally + base-calling hot path on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input: `--batch` (default 8)
independent BAMs of BASELINE configs[1] (1M reads x 29 903 bp each) whose reads are already resident in
HBM as ONE read set (BAM b at positions shifted by b * 29 952), so one HIP tally launch and one HIP call
launch process the batch; the call kernel stores the call records (3 bytes / position) in pinned host
memory and zeroes the count matrix behind itself; native host threads walk each BAM's records to its
consensus sequence (the FASTA content).  Metric: reference positions per second (BASELINE.json), whole job.
`--also-single` measures the same BAMs one per launch afterwards ("single_bam_per_launch"); `--batch 1`
makes that the primary measurement.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--reads R] [--no-cpu-baseline]

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; every rank
processes its own independent BAMs (BASELINE config 4: many-BAM shard, no data-path
collective; weak scaling) and the only collective is the timing barrier / max-reduce.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy


def cpu_baseline(reads, L, mincov, orfs):
    """The oracle timed on this box's host cores (1 thread): the scalar C restatement on the
    full workload, and the Python-loop restatement (the reference's own structure: per-token
    Python loop + O(L^2) ORF pass) on a small sample."""
    import numpy as np
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    c_oracle.tally(reads, L)                                  # warm (page-in)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 8.0:
        counts = c_oracle.tally(reads, L)
        c_oracle.call(counts, mincov, True)
        reps += 1
        if reps >= 40:
            break
    c_s = (time.perf_counter() - t0) / reps
    # Python-loop port on the first 3 000 reads (about 450 k pileup tokens)
    n_s = min(3000, int(reads["n_reads"]))
    nb = len(reads["seq"]) // int(reads["n_reads"])
    sub = {"n_reads": n_s, "pos": reads["pos"][:n_s], "flag": reads["flag"][:n_s], "l_qseq": reads["l_qseq"][:n_s],
           "cigar_off": reads["cigar_off"][:n_s + 1], "cigar": reads["cigar"], "seq_off": reads["seq_off"][:n_s + 1],
           "seq": reads["seq"][:n_s * nb]}
    t0 = time.perf_counter()
    cols = orc.pileup_columns(sub)
    for toks in cols.values():
        orc.tally_tokens(toks)
    py_tok_s = (time.perf_counter() - t0) / max(1, sum(len(v) for v in cols.values()))
    t0 = time.perf_counter()
    Ls = 3000
    orc.build_consensus(mincov, counts[:Ls].astype(np.int64), [{"start": 266, "end": 2900, "strand": "+"}], True, None, True)
    py_walk_s = time.perf_counter() - t0
    tokens = float(np.sum(reads["l_qseq"]))
    return {"value": L / c_s, "unit": "positions/s", "cores": 1, "kind": "port",
            "sample": "oracle/tally_oracle.c (scalar C, -O2) tally+call over the full %d-read workload, %d repetitions"
                      % (int(reads["n_reads"]), reps),
            "seconds_per_bam": c_s,
            "python_port": {"tally_seconds_per_bam_extrapolated": py_tok_s * tokens,
                            "ns_per_token": py_tok_s * 1e9,
                            "walk_seconds_first_3000_positions": py_walk_s,
                            "sample": "oracle/tc_oracle.py per-token loop on the first %d reads; literal O(L^2) "
                                      "BuildConsensus restatement on 3 000 positions" % n_s}}


def run_split_bam(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs):
    """BASELINE configs[4]: one BAM split into `world` contiguous read ranges (genome tiles)."""
    import numpy as np
    from trueconsense_amd import _ffi
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.engine import Context, Walker
    L = len(ref)
    ld = (L + 255) // 256 * 256
    tile = L - 150 + 1
    reads = sy.make_reads(ref, a.reads, seed=7000 + rank, start_range=(tile * rank // world, tile * (rank + 1) // world))
    ctx = Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)     # tally, collective and call on torch's stream
    rs = ctx.upload(reads)
    counts = torch.zeros((7, ld), dtype=torch.int32, device="cuda")
    rec = torch.zeros((3, ld), dtype=torch.uint8, device="cuda")
    walker = Walker([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))

    def step():
        counts.zero_()
        ctx.tally_dev(rs, L, ld, counts.data_ptr(), zero=False)
        td.allreduce_counts(counts)                              # ONE exchange: int32 sum of 7 x ld over the ranks
        ctx.call_dev(counts.data_ptr(), L, ld, a.mincov, True, rec[0].data_ptr(), rec[1].data_ptr(), rec[2].data_ptr())
        if rank == 0:
            h = rec.cpu().numpy()
            return walker(h[0, :L], h[1, :L], h[2, :L])[0]
        return None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ctx.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cons = step()
    fence()
    dt = time.perf_counter() - t0
    tally_ms, tally_n = ctx.profile_get(_ffi.K_TALLY)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        tally_us = 1e3 * tally_ms / max(1, tally_n)
        alg = rs.algorithmic_bytes + 28 * L
        achieved = alg / (tally_us * 1e-6) / 1e9 if tally_us > 0 else 0.0
        total_cov = int(counts[0, :L].sum().item())
        print(json.dumps({
            "metric": "reference positions/sec (ONE BAM of %d reads split over %d GPU(s), BAM -> consensus)" % (a.reads * world, world),
            "value": L * a.steps / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: 29 903-bp reference, one BAM of %d x %d synthetic 150-bp reads, rank r holds "
                                   "the contiguous read range of genome tile r; per step: tally, ONE all-reduce (sum) of the int32 "
                                   "[7][%d] matrix (%d bytes), call kernel, walk on rank 0" % (world, a.reads, ld, 28 * ld),
                       "collective": "gloo (rehearsal on one GPU)" if rehearse else ("RCCL all_reduce" if world > 1 else "none")},
            "roofline": {"kernel": "tally_fast_kernel" if "fast_format=1" in a.ctx_option else "tally_planes_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_us": tally_us},
            "consensus_len": len(cons), "coverage_sum": total_cov, "coverage_sum_expected": 150 * a.reads * world}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--bams", type=int, default=2, help="distinct resident read sets (each --batch BAMs) cycled through per rank")
    ap.add_argument("--mincov", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpu-only", action="store_true", help="leave the host consensus walk out of the step")
    ap.add_argument("--walkers", type=int, default=0, help="host threads for the consensus walks (0 = auto: min(8, cores/ranks))")
    ap.add_argument("--serial", action="store_true", help="no overlap: finish each BAM (GPU + walk) before starting the next")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events (diagnostic)")
    ap.add_argument("--profile-every", type=int, default=8, help="with kernel events on, every n-th step per workspace is launched directly and bracketed with HIP events")
    ap.add_argument("--indels", action="store_true", help="BASELINE configs[2]: indel carriers at CDS boundaries (general kernel + insert sweep)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="KEY=INT", help="tcmi_ctx_set_option on every workspace (diagnostic)")
    ap.add_argument("--split-bam", action="store_true",
                    help="BASELINE configs[4]: ONE BAM of gpus x --reads reads, each rank tallies its contiguous read range, "
                         "one all-reduce (RCCL) of the count matrix per step, base calling on every rank")
    ap.add_argument("--also-single", action="store_true", help="afterwards also measure the same BAMs one per launch (adds a second launch shape)")
    ap.add_argument("--batch", type=int, default=8, help="BAMs per step and launch: their reads are uploaded as one read set at shifted positions")
    ap.add_argument("--slots", type=int, default=12, help="workspaces of the native pipeline (steps queued ahead)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    rehearse = os.environ.get("TCMI_BENCH_REHEARSE") == "1"      # several ranks on ONE GPU over gloo (RCCL wants a GPU per rank)
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from trueconsense_amd import _ffi
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.engine import Pipeline, Walker

    ref, orfs = sy.make_reference()
    L = len(ref)
    if a.split_bam:
        return run_split_bam(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs)
    # (configs[2] is bound by the host's insert-token sweeps: give it more walker threads)
    n_walkers = a.walkers or max(1, min(14 if a.indels else 8, (os.cpu_count() or 1) // max(1, world)))
    pipe = Pipeline(local_rank, slots=a.slots, walkers=n_walkers)   # one stream, `slots` workspaces, native walker threads
    ctx = pipe.ctx
    ctxs = [pipe.slot_context(k) for k in range(a.slots)]
    for kv in a.ctx_option:
        k, v = kv.split("=")
        for c in ctxs:
            c.set_option(k, int(v))
    readsets, host_reads0, all_reads, group0 = [], None, [], []
    B = max(1, a.batch)
    pos_stride = (L + 255) // 256 * 256
    # synthetic BAMs: seeded, generated on a few host threads (numpy releases the GIL), one group of B at a time
    from concurrent.futures import ThreadPoolExecutor
    gen_threads = max(1, min(8, B, (os.cpu_count() or 1) // max(1, world)))
    sites = sy.default_indel_sites(orfs) if a.indels else None
    with ThreadPoolExecutor(gen_threads) as ex:
        for b in range(a.bams):
            group = list(ex.map(lambda k: sy.make_reads(ref, a.reads, seed=1000 * rank + b * B + k + 1, indel_sites=sites), range(B)))
            if host_reads0 is None:
                host_reads0 = group[0]
            if b == 0 and a.also_single:
                group0.extend(group)                            # (kept on the host only when they are needed again)
            all_reads.extend(group if a.indels else [None] * B)
            readsets.append(ctx.upload(group[0]) if B == 1 else ctx.upload_batch(group, pos_stride))
            del group
    alg_reads = readsets[0].algorithmic_bytes                   # 91 B per 150M read
    alg_tally = alg_reads + 28 * L * B                          # + one write of the [L,7] int32 matrix per BAM
    pipe.set_orfs([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    walker = Walker([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))

    def measure(rsets, B, n_steps, n_warm, hreads):
        """n_steps steps of B BAMs each over the read sets `rsets`; -> (seconds, last consensus, kernel times)."""
        def run(n):
            if n <= 0:
                return None
            if a.serial or a.gpu_only:
                cons = None
                for i in range(n):
                    Lg = L if B == 1 else B * pos_stride
                    plain, alt, flags, _ = ctx.step(rsets[i % len(rsets)], Lg, a.mincov, True, want_counts=False)
                    if not a.gpu_only:
                        for k in range(B):
                            o = k * pos_stride
                            cons = walker(plain[o:o + L], alt[o:o + L], flags[o:o + L])[0]
                return cons
            hr = None
            if a.indels:
                hr = [hreads[(i % len(rsets)) * B + k] for i in range(n) for k in range(B)]
            # every consensus is written by the walkers into one preallocated host buffer (allocated and touched
            # outside the timed region); no per-item Python objects inside it
            stride = L + 1 + (4096 if a.indels else 64)
            buf = out_buf[:n * B * stride]
            out, lens, _ = pipe.run([rsets[i % len(rsets)] for i in range(n)], L, a.mincov, True,
                                    host_reads=hr, extra=stride - L - 1, batch=B, pos_stride=pos_stride, out=buf)
            last = n * B - 1
            return out[last * stride:last * stride + int(lens[last])].tobytes()

        out_buf = None
        if not (a.serial or a.gpu_only):
            out_buf = np.empty(max(n_steps, n_warm) * B * (L + 1 + (4096 if a.indels else 64)), np.uint8)
            out_buf.fill(0)                                     # touch every page now

        def fence():
            ctx.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        run(n_warm)
        for c in ([] if a.no_kernel_events else ctxs):
            c.set_option("profile_every", a.profile_every)
            c.profile(True)                                     # HIP events around the kernels, on the pipeline's stream
        fence()
        t0 = time.perf_counter()
        cons = run(n_steps)
        fence()
        dt = time.perf_counter() - t0
        k = {"tally_ms": 0.0, "tally_n": 0, "call_ms": 0.0, "call_n": 0, "gen_ms": 0.0, "gen_n": 0}
        for c in ctxs:
            for name, kid in (("tally", _ffi.K_TALLY), ("call", _ffi.K_CALL), ("gen", _ffi.K_TALLY_GENERAL)):
                m, n = c.profile_get(kid)
                k[name + "_ms"] += m
                k[name + "_n"] += n
            c.profile(False)
        return dt, cons, k

    dt, cons, k = measure(readsets, B, a.steps, a.warmup, all_reads)
    tally_ms, tally_n, call_ms, call_n, gen_ms, gen_n, zero_ms = (k["tally_ms"], k["tally_n"], k["call_ms"], k["call_n"],
                                                                  k["gen_ms"], k["gen_n"], 0.0)
    single = None
    if a.also_single and world == 1 and B > 1 and not a.indels and not (a.serial or a.gpu_only):
        # the same BAMs one per launch (plain BASELINE configs[1] shape), for comparison
        srs = [ctx.upload(r) for r in group0]
        dt1, _, k1 = measure(srs, 1, max(100, a.steps), a.warmup, None)
        us1 = 1e3 * k1["tally_ms"] / max(1, k1["tally_n"])
        alg1 = srs[0].algorithmic_bytes + 28 * L
        single = {"value": L * max(100, a.steps) / dt1, "unit": "positions/s", "ms_per_step": 1e3 * dt1 / max(100, a.steps),
                  "tally_us": us1, "roofline_frac": (alg1 / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS) if us1 > 0 else None,
                  "algorithmic_bytes_per_launch": alg1}
        for r in srs:
            r.free()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        tally_us = 1e3 * tally_ms / max(1, tally_n)
        achieved = alg_tally / (tally_us * 1e-6) / 1e9 if tally_us > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")      # PMC passes are separate runs (see profiles/README)
        if os.path.exists(tp):
            try:
                key = "nibble_tally_hbm_bytes_per_launch" if "fast_format=1" in a.ctx_option else "tally_hbm_bytes_per_launch"
                traffic = json.load(open(tp)).get(key) * B      # measured per 1M-read BAM
            except Exception:
                traffic = None
        out = {
            "metric": "reference positions/sec (1M reads x 29 903 bp per BAM, reads resident in HBM, BAM -> consensus)",
            "value": L * a.steps * B * world / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "bams_per_min": 60.0 * a.steps * B * world / dt,
            "config": {"workload": "BASELINE configs[1]: 29 903-bp reference, %d synthetic 150-bp reads per BAM "
                                   "(~%dx coverage), %d distinct BAMs resident per GPU, %d BAM(s) per step and launch%s"
                                   % (a.reads, a.reads * 150 // L, a.bams * B, B,
                                      "; many-BAM shard over %d GPUs, no data-path collective" % world if world > 1 else ""),
                       "positions": L, "reads_per_bam": a.reads, "mincov": a.mincov,
                       "step": ("tally kernel + call kernel (zeroes the matrix behind itself, stores its records in pinned host memory)"
                                if (a.serial or a.gpu_only or "defer_call=0" in a.ctx_option) else
                                "ONE launch: the tally kernel, whose first blocks also call the matrix the previous step finished "
                                "(position-wise call, records stored in pinned host memory, matrix left zeroed); the last step's call is launched on its own")
                               + ("" if a.gpu_only else " + host consensus walk per BAM"),
                       "overlap": "serial (Python loop)" if (a.serial or a.gpu_only) else
                                  "native pipeline: one stream, %d workspaces queued ahead; walks on %d host threads" % (a.slots, n_walkers)},
            "kernels_us": {"tally_general": 1e3 * gen_ms / max(1, gen_n), "tally": tally_us, "call": 1e3 * call_ms / max(1, call_n), "zero": 1e3 * zero_ms / max(1, tally_n)},
            "roofline": {"kernel": "tally_fast_kernel" if "fast_format=1" in a.ctx_option else "tally_planes_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_tally, "avg_launch_us": tally_us},
        }
        if not a.gpu_only:
            out["consensus_len"] = len(cons) if cons is not None else 0
        if single is not None:
            out["single_bam_per_launch"] = single
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host_reads0, L, a.mincov, orfs)
            out["cpu_baseline"]["cores_on_box"] = os.cpu_count()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
