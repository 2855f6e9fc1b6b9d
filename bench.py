#!/usr/bin/env python3
"""bench.py — BAM file -> consensus FASTA throughput of the pileup-tally + base-calling path on MI355X.

Headline (`value`): reference positions per second, whole job, from the BAM files' COMPRESSED BYTES RESIDENT IN HBM when the clock
starts (tcmi_bamfile_to_device; "inputs already resident in HBM when the timed region starts") to FASTA text on the host: a
"step" is one BAM of BASELINE configs[1] (1M synthetic 150-bp reads over a 29 903-bp reference) through HIP BGZF inflate + CRC +
record chain + pack -> HIP tally + call -> host consensus walk -> FASTA text, with the stages of consecutive BAMs overlapped
(trueconsense_amd.engine.FileRunner.run_resident).  The synthetic BAM files are written, read and copied to the device before the clock
starts.  `--steps K --warmup W`: W untimed BAMs, then the K files as one queue, cycled `repeats` times so that the timed region lasts
>= `--min-seconds` (K x repeats timed BAMs; `--min-seconds 0`: exactly K).

Secondary blocks of the same JSON line:
  file_to_fasta    the PCIe-inclusive rate, measured the same way in the same run: the same files from the page cache (read into pinned
                   memory, H2D of the compressed bytes, then the headline's stages) — round 2's headline
  e2e_single_bam   one BAM file at a time, nothing overlapped: per-stage latency (configs[1] as written)
  cold_kernels, cold_kernels_pipelined   HIP-event times of every kernel, alone on the GPU and in the timed run
  roofline, roofline_hot_path            the dominant kernels against the 8 TB/s HBM roofline (+ `issue`: what does bound them)
  hard_bam, real_bam   files that compress 6 : 1 and 2.5 : 1 (as real data does), one at a time and overlapped, FASTA checked
  cli_batch        the command line with --batch: all four output files per sample
  resident         reads already packed in HBM, 8 BAMs per launch: the tally kernel's rate (what round 1 called the headline)
  cpu_baseline     the same stages on this box's host cores: C restatement of decode + tally + call (oracle/),
                   1 thread and N threads, plus the host walk
  fasta_bit_exact, fasta_all_timed       file 0 / every consensus of the timed path compared with the oracle chain (outside the clock)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--files F] [--indels] [--split-bam [--from-file]]

N > 1: one rank per GPU — launched by the driver with torch.distributed.run, or, when no launcher's WORLD_SIZE is in the environment,
by this script itself (spawn_ranks: N fresh child processes before anything touches a GPU); every rank processes its own BAM files
(BASELINE configs[3]: many-BAM shard, no data-path collective; weak scaling); the only collectives are the timing
barrier and the max-reduce of the elapsed time.  `--split-bam` measures configs[4] instead (one BAM over N ranks, one reduce of
the count matrix per step; `--from-file`: ONE file, every rank decodes its range of the file's BGZF blocks).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy


def KERNELS(_ffi):
    """The kernels of the file -> counts path, by the ids their HIP events are filed under (include/tcmi.h)."""
    return (("inflate_symbols", _ffi.K_INFLATE), ("inflate_copy", _ffi.K_INFLATE_COPY), ("crc32", _ffi.K_CRC), ("records", _ffi.K_RECORDS),
            ("pack_classify", _ffi.K_PACK_CLASSIFY), ("pack", _ffi.K_PACK), ("tally", _ffi.K_TALLY), ("call", _ffi.K_CALL))


def kernel_times(contexts, _ffi, n):
    """-> {kernel: {us_per_bam, launches}} + "inflate" = bgzf_symbols + bgzf_copy (the two kernels of the device inflate)."""
    out = {}
    for name, kid in KERNELS(_ffi):
        m = k = 0
        for c in contexts:
            mm, kk = c.profile_get(kid)
            m, k = m + mm, k + kk
        out[name] = {"us_per_bam": 1e3 * m / max(1, n), "launches": k}
    out["inflate"] = {"us_per_bam": out["inflate_symbols"]["us_per_bam"] + out["inflate_copy"]["us_per_bam"],
                      "launches": out["inflate_symbols"]["launches"] + out["inflate_copy"]["launches"]}
    return out


# ----------------------------------------------------------------------------------------------- inputs
def like_real_data(np, kind, n, seed=1):
    """Qualities and names that make a synthetic BAM compress like real data.  "hard": Illumina-style names (all distinct), qualities drawn
    from four bins — 6 : 1 instead of 35 : 1; "real": qualities drawn like an Illumina run's (a peak at Q36, a tail down to Q2) — 2.5 : 1."""
    rng = np.random.default_rng(seed)
    if kind == "real":
        w = np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004
        qual = rng.choice(np.arange(2, 42, dtype=np.uint8), size=(n, 150), p=w / w.sum())
    else:
        qual = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=(n, 150), p=[0.02, 0.05, 0.13, 0.80])

    def digits(v, w):
        return ((v[:, None] // 10 ** np.arange(w - 1, -1, -1)[None, :]) % 10 + 48).astype(np.uint8)
    lit = lambda t: np.tile(np.frombuffer(t, np.uint8), (n, 1))
    names = np.concatenate([lit(b"A00123:45:HXXXXX:"), digits(rng.integers(1, 5, n), 1), lit(b":"), digits(rng.integers(1101, 2679, n), 4), lit(b":"),
                            digits(rng.integers(1000, 33000, n), 5), lit(b":"), digits(rng.integers(1000, 37000, n), 5)], axis=1)
    return qual, names


def write_inputs(tmp, ref, orfs, n_files, n_reads, rank, indels, level, kind="headline"):
    """Seeded synthetic BAM files of configs[1] (or [2] with --indels; kind "hard" / "real": files that compress like real data, for
    counter passes over them).  -> (paths, reads of file 0)."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    sites = sy.default_indel_sites(orfs) if indels else None
    first = {}

    def one(k):
        reads = sy.make_reads(ref, n_reads, seed=1000 * rank + k + 1, indel_sites=sites)
        path = os.path.join(tmp, "r%d_%d.bam" % (rank, k))
        if indels:
            bamwriter.write_bam(path, reads, "MN908947.3", len(ref), level=level)
        elif kind != "headline":
            qual, names = like_real_data(np, kind, n_reads, seed=k + 1)
            bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n_reads, -1), 150, "MN908947.3", len(ref), level=level,
                                     qual=qual, names=names)
            reads["qual"] = qual.reshape(-1)                      # (the oracle chain's region pile-ups read the qualities)
        else:
            bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n_reads, -1), 150,
                                     "MN908947.3", len(ref), level=level)
        if k == 0:
            first["reads"] = reads
        return path

    with ThreadPoolExecutor(max(1, min(8, n_files, (os.cpu_count() or 1)))) as ex:
        paths = list(ex.map(one, range(n_files)))
    return paths, first["reads"]


def inflated_size(path):
    """Sum of the ISIZE fields of a BGZF file's blocks (4 bytes at the end of every gzip member)."""
    import struct
    raw = open(path, "rb").read()
    off = tot = 0
    while off + 18 <= len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        tot += struct.unpack_from("<I", raw, off + bsize - 4)[0]
        off += bsize
    return tot


# ----------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(paths, L, mincov, orfs, n_threads, budget_s=12.0):
    """The same stages on the host: oracle/bam_oracle.c (sequential gunzip + record walk), oracle/tally_oracle.c
    (scalar tally + call) and the product's own host walk (it is host code on both sides), per BAM file.
    Once on one thread, then `n_threads` threads each working through the files for about `budget_s` seconds."""
    import threading
    import numpy as np
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    from trueconsense_amd.engine import Walker
    walker = Walker([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))

    def one(path, stages=None):
        t0 = time.perf_counter()
        reads = c_oracle.read_bam(path)
        t1 = time.perf_counter()
        counts = c_oracle.tally(reads, max(L, c_oracle.extent(reads, L)))
        t2 = time.perf_counter()
        plain, alt, flags = c_oracle.call(counts, mincov, True)
        cons = walker(plain[:L], alt[:L], flags[:L])[0]
        t3 = time.perf_counter()
        if stages is not None:
            stages.append((t1 - t0, t2 - t1, t3 - t2))
        return cons

    one(paths[0])                                                # warm (page cache, lazy binding)
    st = []
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or (time.perf_counter() - t0 < 4.0 and reps < 8):
        one(paths[reps % len(paths)], st)
        reps += 1
    s1 = (time.perf_counter() - t0) / reps
    done = [0] * n_threads
    stop = time.perf_counter() + budget_s

    def work(t):
        k = t
        while time.perf_counter() < stop:
            one(paths[k % len(paths)])
            k += n_threads
            done[t] += 1

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    tn0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dtn = time.perf_counter() - tn0
    total = sum(done)
    # the reference's own loop structure (per-token Python loop) on a small sample, for scale
    reads = c_oracle.read_bam(paths[0])
    n_s = min(3000, int(reads["n_reads"]))
    sub = {"n_reads": n_s, "pos": reads["pos"][:n_s], "flag": reads["flag"][:n_s], "l_qseq": reads["l_qseq"][:n_s],
           "cigar_off": reads["cigar_off"][:n_s + 1], "cigar": reads["cigar"], "seq_off": reads["seq_off"][:n_s + 1],
           "seq": reads["seq"]}
    tp = time.perf_counter()
    cols = orc.pileup_columns(sub)
    for toks in cols.values():
        orc.tally_tokens(toks)
    py_tok_s = (time.perf_counter() - tp) / max(1, sum(len(v) for v in cols.values()))
    return {"value": L * total / dtn, "unit": "positions/s", "cores": n_threads, "kind": "port",
            "sample": "BAM file -> consensus per file: oracle/bam_oracle.c (gunzip + records) + oracle/tally_oracle.c "
                      "(scalar tally + call, -O2) + host walk; %d threads each looping over the %d bench files for %.0f s "
                      "(%d BAMs done)" % (n_threads, len(paths), budget_s, total),
            "seconds_per_bam_per_thread": dtn * n_threads / max(1, total),
            "one_thread": {"value": L / s1, "unit": "positions/s", "cores": 1, "seconds_per_bam": s1, "bams": reps,
                           "stage_seconds": {"decode": float(np.mean([a for a, _, _ in st])),
                                             "tally": float(np.mean([b for _, b, _ in st])),
                                             "call_walk": float(np.mean([c for _, _, c in st]))}},
            "python_loop_ns_per_token": py_tok_s * 1e9,
            "python_loop_tally_seconds_per_bam_extrapolated": py_tok_s * float(np.sum(reads["l_qseq"])),
            "cores_on_box": os.cpu_count()}


# ----------------------------------------------------------------------------------------------- configs[4]
def run_split_bam(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs):
    """BASELINE configs[4]: one BAM split into `world` contiguous read ranges (genome tiles); per step every rank
    tallies its range, ONE reduce (sum) of the int32 [7][ld] matrix to rank 0, which calls and walks."""
    from trueconsense_amd import _ffi
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.engine import Context, Walker
    L = len(ref)
    ld = (L + 255) // 256 * 256
    tile = L - 150 + 1
    ctx = Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)     # tally, collective and call on torch's stream
    if a.from_file:
        return run_split_bamfile(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs, ctx, tile)
    reads = sy.make_reads(ref, a.reads, seed=7000 + rank, start_range=(tile * rank // world, tile * (rank + 1) // world))
    rs = ctx.upload(reads)
    counts = torch.zeros((7, ld), dtype=torch.int32, device="cuda")
    rec = torch.zeros((3, ld), dtype=torch.uint8).pin_memory()                     # the call kernel stores across PCIe
    walker = Walker([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    rec_np = rec.numpy()

    def step():
        counts.zero_()
        ctx.tally_dev(rs, L, ld, counts.data_ptr(), zero=False)
        td.reduce_counts(counts, dst=0)                              # ONE exchange: int32 sum of 7 x ld, to rank 0 only
        if rank == 0:
            ctx.call_dev(counts.data_ptr(), L, ld, a.mincov, True, rec[0].data_ptr(), rec[1].data_ptr(), rec[2].data_ptr())
            torch.cuda.current_stream().synchronize()
            return walker(rec_np[0, :L], rec_np[1, :L], rec_np[2, :L])[0]
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
        real = rs.device_bytes + 28 * L
        alg = rs.algorithmic_bytes + 28 * L
        achieved = real / (tally_us * 1e-6) / 1e9 if tally_us > 0 else 0.0
        total_cov = int(counts[0, :L].sum().item())
        print(json.dumps({
            "metric": "reference positions/sec (ONE BAM of %d reads split over %d GPU(s), reads resident in HBM -> consensus)" % (a.reads * world, world),
            "value": L * a.steps / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: 29 903-bp reference, one BAM of %d x %d synthetic 150-bp reads, rank r holds "
                                   "the contiguous read range of genome tile r; per step: tally, ONE reduce (sum) of the int32 "
                                   "[7][%d] matrix (%d bytes) to rank 0, call kernel + walk on rank 0" % (world, a.reads, ld, 28 * ld),
                       "collective": "gloo (rehearsal on one GPU)" if rehearse else ("RCCL reduce" if world > 1 else "none")},
            "roofline": {"kernel": "tally_planes_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": real,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_us": tally_us},
            "consensus_len": len(cons), "coverage_sum": total_cov, "coverage_sum_expected": 150 * a.reads * world}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_split_bamfile(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs, ctx, tile):
    """configs[4] from ONE FILE: rank 0 writes one BAM of world x --reads reads (outside the clock); every rank maps it and, per
    step, runs the product's function for this shape (distributed.consensus_split_bamfile): tcmi_split_step in C — the compressed bytes
    of ITS contiguous range of BGZF blocks to its GPU, inflate, index and pack the records that start there, tally, ONE reduce (sum) of
    the int32 [7][ld] matrix to rank 0 through the RCCL hook, call kernel on rank 0 —, then the insert candidates' entries to rank 0,
    vote, walk, FASTA text.  Nothing of the file is decoded on the host, and no rank decodes another rank's blocks (one block at a
    range's end excepted: the last record may run into it)."""
    from trueconsense_amd import _ffi
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.engine import DeviceBam
    from trueconsense_amd.io import bamwriter
    import numpy as np
    L = len(ref)
    ld = (L + 255) // 256 * 256
    tmp = a.tmp or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    path = os.path.join(tmp, "tcmi_split_%s.bam" % os.environ.get("MASTER_PORT", str(os.getpid())))
    t_gen = time.perf_counter()
    want_counts = None
    if rank == 0:
        from oracle import c_oracle
        want_counts = np.zeros((L, 7), np.int64)
        for r in range(world):                                       # tile by tile: the file is coordinate-sorted
            reads = sy.make_reads(ref, a.reads, seed=7000 + r, start_range=(tile * r // world, tile * (r + 1) // world))
            bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(a.reads, -1), 150, "MN908947.3", L, level=a.level,
                                     part=(r == 0, r == world - 1), first_id=r * a.reads)
            want_counts += c_oracle.tally(reads, L)                  # (the checker: outside the clock)
            del reads
    t_gen = time.perf_counter() - t_gen
    if dist is not None:
        dist.barrier()
    d = DeviceBam(path)
    first, count = td.block_range(d.n_blocks, rank, world)
    rows = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs]
    # The exchange: tcmi_split_step's C hook over an RCCL communicator of this job's own (include/tcmi_rccl.h: ncclReduce queued on the
    # context's stream behind the tally) — at ONE rank too, so that the collective itself runs wherever this leg runs; a rehearsal of
    # several ranks on one GPU (gloo) keeps torch.distributed's reduce.
    # (distributed.split_reduce_hook: the same call the command line's split worker makes — all ranks get the same answer)
    comm, user, hook_kind = td.split_reduce_hook(rank, world, rccl=not rehearse)
    if user is None:
        hook_kind += " (gloo, rehearsal on one GPU)" if rehearse else (" (RCCL)" if world > 1 else " (no-op at one rank)")
    rs0 = ctx.upload_bamfile(d, blocks=(first, count))               # (outside the clock: how many records start in this rank's range)
    n_mine = int(rs0.n_reads)
    rs0.free()
    tms = []

    def step():
        # the product's function for this shape, all the way to the FASTA text: tcmi_split_step (H2D of the range's compressed bytes,
        # inflate, index, pack, tally, the reduce, call on rank 0 — in C), the insert candidates' entries gathered to rank 0, vote, walk
        tm = {}
        text = td.consensus_split_bamfile(path, L, rows, a.mincov, True, "S", rank, world, device=local_rank, rccl_user=user, ctx=ctx, dbam=d, timings=tm)
        tms.append(tm)
        return text

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    del tms[:]
    # (the interpreter's cyclic collector is held off for the timed steps: with torch imported a full collection takes 50 - 100 ms, and one
    #  of them landing in a 55-ms timed region — it did, in two calls of three — is a property of this script, not of the step)
    import gc
    gc.collect()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cons = step()
    fence()
    dt_mine = time.perf_counter() - t0
    gc.enable()
    dt = dt_mine
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, (dt_mine, n_mine, int(count)))
    else:
        per_rank = [(dt_mine, n_mine, int(count))]
    # ... and once more outside the clock, with the count matrix handed back, for the checks
    parts = td.consensus_split_bamfile(path, L, rows, a.mincov, True, "S", rank, world, device=local_rank, rccl_user=user, ctx=ctx, dbam=d, return_parts=True)
    if rank == 0:
        from oracle import tc_oracle as orc
        got = parts[1].astype(np.int64)
        has, ins = orc.list_inserts(want_counts, a.mincov, lambda pos1: [])
        want, _ = orc.build_consensus(a.mincov, want_counts, [dict(o) for o in orfs], True, ins if has else None, True)
        want_text = orc.fasta_text("S", a.mincov, want)
        step_ms = 1e3 * float(np.mean([t_["step"] for t_ in tms]))
        print(json.dumps({
            "consensus_split_bamfile_fasta_exact": bool(parts[0] == want_text),
            "metric": "reference positions/sec (ONE BAM FILE of %d reads, %d GPU(s) each decoding its range of the file's BGZF blocks -> consensus FASTA)" % (a.reads * world, world),
            "value": L * a.steps / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "ms_per_step_per_rank": [1e3 * x[0] / a.steps for x in per_rank],
            "ms_per_step_in_tcmi_split_step": step_ms, "ms_per_step_verdicts_entries_vote": 1e3 * float(np.mean([t_.get("entries", 0.0) for t_ in tms])),
            "ms_per_step_release": 1e3 * float(np.mean([t_.get("release", 0.0) for t_ in tms])), "ms_per_step_in_the_function": 1e3 * float(np.mean([t_.get("total", 0.0) for t_ in tms])),
            "ms_per_step_median": 1e3 * float(np.median([t_.get("total", 0.0) for t_ in tms])),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: 29 903-bp reference, ONE BAM file of %d x %d synthetic 150-bp reads (%d bytes, %d BGZF blocks, "
                                   "zlib level %d); a step is distributed.consensus_split_bamfile — the product's function: tcmi_split_step in C (every rank "
                                   "sends the compressed bytes of its contiguous block range to its GPU, inflates / indexes / packs / tallies there, ONE reduce (sum) "
                                   "of the int32 [7][%d] matrix + range table (%d bytes) to rank 0 through the hook, call kernel on rank 0), the ranks' verdicts, "
                                   "the insert candidates' entries to rank 0, vote, walk, FASTA text" % (world, a.reads, d.file_bytes, d.n_blocks, a.level, ld, 4 * (7 * ld + 6 * world + 1)),
                       "collective": hook_kind,
                       "blocks_per_rank": [x[2] for x in per_rank], "reads_per_rank": [x[1] for x in per_rank],
                       "input_generation_seconds_outside_clock": t_gen},
            "counts_bit_exact": bool(np.array_equal(got, want_counts)), "fasta_bit_exact": bool(cons == want_text),
            "consensus_len": len(want), "coverage_sum": int(got[:, 0].sum()), "coverage_sum_expected": 150 * a.reads * world}))
    td.split_reduce_hook_close(comm)
    d.close()
    if dist is not None:
        dist.barrier()
        if rank == 0:
            os.remove(path)
        dist.destroy_process_group()
    elif rank == 0:
        os.remove(path)


# ----------------------------------------------------------------------------------------------- --gpus N without a launcher
def spawn_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run around it: this process — BEFORE it imports torch or touches a GPU, and
    never by re-executing itself — starts N fresh children, one rank per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment (as the driver's launcher and TrueConsense.run_gpus do), lets them write to its own stdout / stderr
    (rank 0 prints the ONE JSON line), ends the others when one fails, and returns the worst exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env0, RANK=str(k), LOCAL_RANK=str(k))) for k in range(n)]
    rcs = [None] * n
    failed_at = None
    try:
        while any(r is None for r in rcs):
            for k, p_ in enumerate(procs):
                if rcs[k] is None:
                    rcs[k] = p_.poll()
                    if rcs[k] not in (None, 0) and failed_at is None:
                        failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > 20.0:      # (the others may sit in a collective waiting for the one that died)
                break
            time.sleep(0.05)
    finally:
        for k, p_ in enumerate(procs):
            if rcs[k] is None:
                p_.terminate()
        for k, p_ in enumerate(procs):
            if rcs[k] is None:
                try:
                    rcs[k] = p_.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p_.kill()
                    rcs[k] = p_.wait()
    return max((abs(r) for r in rcs), default=0)


# ----------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed BAM files (each one BAM file -> FASTA)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the K files are cycled `repeats` times in one queue so that the timed region lasts at least this long (0: exactly K)")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--files", type=int, default=8, help="distinct synthetic BAM files per rank, cycled through")
    ap.add_argument("--level", type=int, default=6, help="zlib level the synthetic BAMs are written with (samtools' default is 6)")
    ap.add_argument("--mincov", type=int, default=30)
    ap.add_argument("--decoders", type=int, default=0, help="BAMs being decoded at a time (0 = auto)")
    ap.add_argument("--decode-threads", type=int, default=0, help="native threads per BAM decode (0 = auto)")
    ap.add_argument("--walkers", type=int, default=2)
    ap.add_argument("--gpu-streams", type=int, default=0, help="contexts (stream + device arena) the GPU stage of consecutive BAMs alternates between (0 = auto: 8, fewer on a host with few cores per rank — a context's thread spins while it waits for its stream)")
    ap.add_argument("--indels", action="store_true", help="BASELINE configs[2]: indel carriers at CDS boundaries")
    ap.add_argument("--file-kind", choices=("headline", "hard", "real"), default="headline",
                    help="the bench files themselves written to compress like real data (6 : 1 / 2.5 : 1): counter passes over those decoders (tools/pmc_e2e.sh)")
    ap.add_argument("--host-decode", action="store_true", help="decode the BAMs on the host (tcmi_bam_load) instead of on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="skip the HBM-resident kernel-rate leg")
    ap.add_argument("--no-cli-batch", action="store_true", help="skip the leg through the command line (--batch, four output files per sample)")
    ap.add_argument("--no-hard-bam", action="store_true", help="skip the leg with the file that compresses like real data")
    ap.add_argument("--no-configs2", action="store_true", help="skip the configs[2] leg (indel carriers) of the default line")
    ap.add_argument("--no-configs0", action="store_true", help="skip the configs[0] leg (10k reads: the Python stand-in for the reference's run beside the command line)")
    ap.add_argument("--only-resident", action="store_true", help="only the HBM-resident leg (short runs under rocprofv3)")
    ap.add_argument("--resident-batch", type=int, default=8, help="BAMs per launch of the resident leg")
    ap.add_argument("--resident-steps", type=int, default=200)
    ap.add_argument("--slots", type=int, default=12, help="workspaces of the native pipeline in the resident leg")
    ap.add_argument("--profile-every", type=int, default=8)
    ap.add_argument("--ctx-option", action="append", default=[], metavar="KEY=INT", help="tcmi_ctx_set_option (diagnostic)")
    ap.add_argument("--tmp", default=None, help="directory for the synthetic BAM files (default: /dev/shm or $TMPDIR)")
    ap.add_argument("--from-file", action="store_true", help="with --split-bam: ONE BAM file, every rank decodes its range of the file's BGZF blocks on its GPU")
    ap.add_argument("--split-bam", action="store_true",
                    help="BASELINE configs[4]: ONE BAM of gpus x --reads reads, each rank tallies its contiguous read range, "
                         "one reduce (RCCL) of the count matrix per step, base calling on rank 0")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:               # no launcher around us: be it (nothing has touched a GPU yet)
        sys.exit(spawn_ranks(a.gpus))
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
    from trueconsense_amd.engine import Context, FileRunner, Pipeline

    ref, orfs = sy.make_reference()
    L = len(ref)
    if a.split_bam:
        return run_split_bam(a, rank, local_rank, world, rehearse, dist, torch, ref, orfs)

    cores = max(1, (os.cpu_count() or 1) // max(1, world if not rehearse else 1))
    cores = min(cores, 16)                                       # the box's CPU share per GPU
    decoders = a.decoders or (3 if cores >= 12 else 2 if cores >= 8 else 1)     # (readers are near their limit at two on slower hosts)
    decode_threads = a.decode_threads or max(1, cores // decoders)
    if a.gpu_streams <= 0:
        a.gpu_streams = min(8, max(3, cores // 2))

    if a.only_resident:
        ctx = Context(local_rank)
        res = resident_leg(a, ctx, Pipeline, local_rank, np, _ffi, sy, ref, orfs, L, rank, lambda: (ctx.sync(), torch.cuda.synchronize()))
        res["resident"].update(metric="reference positions/sec (reads resident in HBM -> consensus)", n_gpus=world)
        print(json.dumps(res))
        return
    # ---- inputs: BAM files on disk, written outside the clock ------------------------------------------
    base = a.tmp or ("/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (4 << 30) else tempfile.gettempdir())
    tmp = tempfile.mkdtemp(prefix="tcmi_bench_", dir=base)
    try:
        t_gen = time.perf_counter()
        paths, reads0 = write_inputs(tmp, ref, orfs, a.files, a.reads, rank, a.indels, a.level, a.file_kind)
        t_gen = time.perf_counter() - t_gen
        out = run_bench(a, rank, local_rank, world, rehearse, dist, torch, np, _ffi, sy, Context, FileRunner, Pipeline,
                        ref, orfs, L, paths, reads0, decoders, decode_threads, cores, t_gen)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_bench(a, rank, local_rank, world, rehearse, dist, torch, np, _ffi, sy, Context, FileRunner, Pipeline, ref, orfs, L,
              paths, reads0, decoders, decode_threads, cores, t_gen):
    t_bench0 = time.perf_counter()
    ctx = Context(local_rank)
    for kv in a.ctx_option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gff_rows = [{"start": o["start"], "end": o["end"], "strand": o["strand"]} for o in orfs]
    file_of = lambda i: paths[i % len(paths)]

    # ---- headline: BAM files -> FASTA text, stages overlapped ------------------------------------------
    runner = FileRunner(ctx, gff_rows, a.mincov, True, decoders=decoders, decode_threads=decode_threads, walkers=a.walkers,
                        gpu_streams=a.gpu_streams)
    runner.device_decode = not a.host_decode
    for kv in a.ctx_option:                                      # (experiments: e.g. verify_crc=0)
        k, v = kv.split("=")
        for c in runner.contexts:
            c.set_option(k, int(v))
    def timed_leg(run):
        """run(indices, names) -> FASTA texts.  Warm-up, one untimed pass over the K files to size the job (the K timed steps take
        ~ 25 ms on one GPU, too short for a stable figure), then ONE queue of K x `repeats` files (>= --min-seconds of work, one
        pipeline fill and drain) as the timed region, bracketed by sync + barrier; the slowest rank's time counts."""
        if a.warmup > 0:
            run(list(range(a.warmup)), None)
        for c in runner.contexts:
            c.profile(True)                                      # HIP events around every kernel of the cold path
        fence()
        t0 = time.perf_counter()
        run(list(range(a.steps)), None)
        fence()
        repeats = max(1, int(np.ceil(a.min_seconds / max(1e-6, time.perf_counter() - t0)))) if a.min_seconds > 0 else 1
        if dist is not None:
            t = torch.tensor([repeats], dtype=torch.int64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
        n_timed = a.steps * repeats
        runner.seconds = {k: 0.0 for k in runner.seconds}
        runner.decoded_on = {k: 0 for k in runner.decoded_on}
        for c in runner.contexts:
            c.profile(False)
            c.profile(True)
        fence()
        t0 = time.perf_counter()
        fastas = run(list(range(n_timed)), ["S%d" % (i % len(paths)) for i in range(n_timed)])
        fence()
        dt = time.perf_counter() - t0
        cold = kernel_times(runner.contexts, _ffi, n_timed)
        for c in runner.contexts:
            c.profile(False)
        dt_ranks = [dt]
        if dist is not None:
            dt_ranks = [None] * world
            dist.all_gather_object(dt_ranks, dt)
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return {"fastas": fastas, "dt": dt, "dt_ranks": dt_ranks, "repeats": repeats, "n_timed": n_timed, "cold": cold,
                "busy": {k: v / n_timed for k, v in runner.seconds.items()}, "decoded_on": dict(runner.decoded_on)}

    # ---- the PCIe-inclusive leg: BAM FILES (page cache) -> FASTA text: read into pinned memory, H2D of the compressed bytes, GPU, walk ----
    leg_file = timed_leg(lambda idx, names: runner.run([file_of(i) for i in idx], names=names, ref_len=L))
    # ---- headline: the same files' compressed bytes RESIDENT IN HBM when the clock starts (read and copied before it) -> FASTA text ----
    leg = leg_file
    dbams = []
    if not a.host_decode:
        from trueconsense_amd.engine import DeviceBam
        dbams = [DeviceBam(p_).to_device(ctx) for p_ in paths]
        leg = timed_leg(lambda idx, names: runner.run_resident([dbams[i % len(dbams)] for i in idx], names=names, ref_len=L))
    fastas, dt, dt_ranks, repeats, n_timed, cold = leg["fastas"], leg["dt"], leg["dt_ranks"], leg["repeats"], leg["n_timed"], leg["cold"]
    if rank != 0:
        if not a.no_resident:
            resident_leg(a, ctx, Pipeline, local_rank, np, _ffi, sy, ref, orfs, L, rank, fence)
        return None

    out = {
        "metric": "reference positions/sec (BAM file -> consensus FASTA, 1M reads x 29 903 bp per BAM)",
        "value_definition": ("compressed BAM file bytes resident in HBM when the clock starts -> FASTA text on the host (BGZF inflate, record chain, "
                             "pack, tally, call on the GPU; consensus walk on the host); the PCIe-inclusive rate from files in the page "
                             "cache is file_to_fasta" if dbams else "BAM files in the page cache -> FASTA text, host decode (--host-decode)"),
        "value": L * n_timed * world / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / n_timed, "ms_per_step_per_rank": {"min": 1e3 * min(dt_ranks) / n_timed, "max": 1e3 * max(dt_ranks) / n_timed},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic", "repeats": repeats, "timed_seconds": dt, "bams_per_min": 60.0 * n_timed * world / dt,
        "config": {"workload": "BASELINE configs[%d]: 29 903-bp reference, %d synthetic 150-bp reads per BAM (~%dx coverage)%s, "
                               "%d distinct BAM files per GPU (zlib level %d, written before the clock starts) cycled through; "
                               "a step = one BAM FILE -> FASTA text%s"
                               % (2 if a.indels else 1, a.reads, a.reads * 150 // L,
                                  ", indel carriers at CDS boundaries" if a.indels else "", len(paths), a.level,
                                  "; many-BAM shard over %d GPUs, no data-path collective" % world if world > 1 else ""),
                   "positions": L, "reads_per_bam": a.reads, "mincov": a.mincov, "file_kind": a.file_kind,
                   "stages": "[file_to_fasta only: read (file bytes into pinned memory, BGZF block table, BAM header; host, %d files in flight) -> "
                             "H2D of the COMPRESSED file ->] HIP: BGZF inflate + record chain + CIGAR projection / "
                             "classification / bit-plane pack -> HIP tally + call (records to pinned host memory) -> host walk + "
                             "FASTA text (%d threads); stages of consecutive BAMs overlap on %d GPU contexts; decoded on: %s"
                             % (decoders, a.walkers, a.gpu_streams, json.dumps(leg["decoded_on"])),
                   "bam_file_bytes": os.path.getsize(paths[0]), "bam_inflated_bytes": inflated_size(paths[0]),
                   "input_generation_seconds_outside_clock": t_gen},
        "e2e_stage_busy_seconds_per_bam": leg["busy"],
        "cold_kernels_pipelined": cold,
    }
    if dbams:
        fb = os.path.getsize(paths[0])
        out["file_to_fasta"] = {
            "value": L * leg_file["n_timed"] * world / leg_file["dt"], "unit": "positions/s", "ms_per_step": 1e3 * leg_file["dt"] / leg_file["n_timed"],
            "repeats": leg_file["repeats"], "timed_seconds": leg_file["dt"], "e2e_stage_busy_seconds_per_bam": leg_file["busy"],
            "decoded_on": leg_file["decoded_on"], "compressed_GB_per_s_over_pcie": fb * leg_file["n_timed"] / leg_file["dt"] / 1e9,
            "note": "PCIe-inclusive: the same files from the page cache — read into pinned memory, H2D of the compressed bytes (%d B per BAM), then the "
                    "headline's stages.  tools/h2d_rate.py measures what a pinned copy of that size alone takes on the box (profiles/README.md): "
                    "this leg runs at that rate" % fb}
        out["file_to_fasta"]["fastas_equal_headline"] = leg_file["fastas"][:len(paths)] == fastas[:len(paths)]

    # ---- one BAM at a time, nothing overlapped ---------------------------------------------------------
    single = FileRunner(ctx, gff_rows, a.mincov, True, decoders=1, decode_threads=min(16, cores), walkers=1, gpu_streams=1)
    single.device_decode = not a.host_decode
    lat = []
    single.run([file_of(0)], ref_len=L)                          # (its context's workspace and arena are allocated on first use)
    for i in range(3):
        single.seconds = {k: 0.0 for k in single.seconds}
        t1 = time.perf_counter()
        single.run([file_of(i)], ref_len=L)
        lat.append((time.perf_counter() - t1, dict(single.seconds)))
    best = min(lat, key=lambda x: x[0])
    out["e2e_single_bam"] = {"seconds": best[0], "positions_per_s": L / best[0], "stage_seconds": best[1],
                             "decode_threads": min(16, cores), "runs": [x[0] for x in lat]}
    # kernel times with nothing else on the GPU (the pipelined run above overlaps the kernels of three BAMs, which stretches
    # every one of them): HIP events on the single runner's stream, 8 BAMs one after the other
    sctx = single.contexts[0]
    sctx.profile(True)
    n_k = 8
    single.run([file_of(i) for i in range(n_k)], ref_len=L)
    cold = kernel_times([sctx], _ffi, n_k)
    sctx.profile(False)
    out["cold_kernels"] = cold
    # north_star's one numeric roofline target (>= 40 % of HBM peak) is about the tally kernel, in the path that ships: one launch per
    # BAM file.  Three byte counts over the SAME launch duration (HIP events, alone on the GPU and in the timed run): what the kernel
    # reads from the device (the packed read set) + the matrix; what the counters saw; SURVEY 8-d's BAM-form bytes.
    if not a.host_decode:
        from trueconsense_amd.engine import DeviceBam as _DB
        d0 = _DB(paths[0])
        rs0 = sctx.upload_bamfile(d0)
        dev_b, alg_b = rs0.device_bytes + 28 * L, rs0.algorithmic_bytes + 28 * L
        rs0.free()
        d0.close()
        us1, usp = cold["tally"]["us_per_bam"], out["cold_kernels_pipelined"]["tally"]["us_per_bam"]
        pmc_b, pmc_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            rk = next((k for k in ("round6", "round5", "round4") if k in tj), None)
            pmc_b = ((tj.get(rk) or {}).get("kernels", {}).get("tally_planes_kernel") or {}).get("hbm_bytes")
            pmc_src = "profiles/traffic.json %s (FETCH_SIZE + WRITE_SIZE of tally_planes_kernel, one 1M-read BAM per launch)" % rk
        except Exception:
            pass
        fr = lambda b_, us_: (b_ / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS) if (b_ and us_ > 0) else None
        out["roofline_tally"] = {
            "kernel": "tally_planes_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "launches": "one per BAM file (the path that ships)",
            "avg_launch_us": us1, "avg_launch_us_pipelined": usp,
            "achieved": dev_b / (us1 * 1e-6) / 1e9 if us1 > 0 else 0.0, "frac": fr(dev_b, us1),
            "device_bytes": {"bytes_per_launch": dev_b, "frac": fr(dev_b, us1), "frac_pipelined": fr(dev_b, usp),
                             "what": "the packed read set (52 B per 150-bp read, read once) + 28 B x positions (the count matrix)"},
            "counter_bytes": {"bytes_per_launch": pmc_b, "frac": fr(pmc_b, us1), "source": pmc_src},
            "survey_8d_bytes": {"bytes_per_launch": alg_b, "frac": fr(alg_b, us1), "frac_pipelined": fr(alg_b, usp),
                                "what": "SURVEY 8-d's algorithmic bytes: 12 + 4 n_cigar + l/2 per read (the BAM-form record fields) + the count matrix"},
            "traffic": pmc_b,
            "note": "alone on the GPU (single stream, HIP events); `resident.roofline` is the eight-BAMs-per-launch leg no file path uses"}

    # ---- the GPU legs first, back to back; the host-side checks of what they produced (a minute of oracle work) come behind them ----
    deferred = []
    # BASELINE configs[2] in the same line: indel carriers at CDS boundaries, insert candidates resolved in the resident stream
    if not a.indels and not a.host_decode and not a.no_configs2:
        out["configs2"] = configs2_leg(a, np, sy, runner, ctx, ref, orfs, L, os.path.dirname(paths[0]), rank, min(8, cores), deferred)
    # a file that compresses like real data (distinct names, binned random qualities): same pipeline, one BAM at a time
    if not a.indels and not a.host_decode and not a.no_hard_bam:
        out["hard_bam"] = hard_bam_leg(a, single, sctx, _ffi, np, sy, ref, L, os.path.dirname(paths[0]), runner, ctx)
        out["real_bam"] = hard_bam_leg(a, single, sctx, _ffi, np, sy, ref, L, os.path.dirname(paths[0]), runner, ctx, kind="real")
    # the command line itself with --batch: all four output files per sample, written by the native runner
    if not a.host_decode and not a.no_cli_batch:
        out["cli_batch"] = cli_batch_leg(a, paths, ref, orfs, L, sy, fastas)
    # secondary: reads resident in HBM, the tally kernel's rate and its roofline
    if not a.no_resident:
        res = resident_leg(a, ctx, Pipeline, local_rank, np, _ffi, sy, ref, orfs, L, rank, fence)
        out["resident"] = res["resident"]
        out["resident"]["roofline"] = res["roofline"]
    out["gpu_legs_done_after_seconds"] = time.perf_counter() - t_bench0

    # ---- bit-exactness of what was timed, against the oracle chain (outside the clock) -----------------
    chain_t = {}
    out["fasta_all_timed"] = check_all_fastas(a, np, paths, fastas, L, a.mincov, orfs, workers=min(8, cores), timing=chain_t)
    out.update({"fasta_bit_exact": out["fasta_all_timed"]["first_mismatch"] is None and len(fastas) > 0,
                "fasta_sha256": hashlib.sha256(fastas[0].encode()).hexdigest()[:16], "consensus_len": len(fastas[0].split("\n")[1])})
    for fn in deferred:
        fn()
    # ---- BASELINE configs[0]: the Python stand-in for the reference's whole run beside the product's command line, one 10k-read file ----
    if not a.indels and not a.host_decode and not a.no_configs0:
        out["configs0"] = configs0_leg(np, sy, ref, orfs, L, os.path.dirname(paths[0]), a.mincov)

    # ---- rooflines of the cold path's kernels (HIP events of the timed run above; traffic from the committed PMC passes) ----
    out.update(cold_rooflines(a, out["cold_kernels_pipelined"], cold, os.path.getsize(paths[0]), out["config"].get("bam_inflated_bytes", 0), reads0))
    if "all_kernels_of_a_bam" in out["roofline"].get("issue", {}):
        ak = out["roofline"]["issue"]["all_kernels_of_a_bam"]
        ak["ms_per_step_pipelined"] = out["ms_per_step"] * world
        ak["issue_floor_over_pipelined_time"] = ak["issue_floor_ms"] / ak["ms_per_step_pipelined"]
        if ak.get("hbm_bytes_per_bam"):                                # what the pipeline pulls through the L2s' far side, all kernels together
            agg = ak["hbm_bytes_per_bam"] / (ak["ms_per_step_pipelined"] * 1e-3) / 1e9
            ak["aggregate_traffic"] = {"achieved": agg, "unit": "GB/s", "frac": agg / HBM_PEAK_GBS}
        out["parked"] = {k: v.get("parked") for k, v in out["roofline"]["issue"]["kernels"].items() if v.get("parked") is not None}
    # (`achieved` divides by the average duration of ONE launch; with several contexts the launches of different BAMs overlap and
    #  each one is stretched by the others.  What the kernel class moves per second of the timed run:)
    for blk in ("roofline", "roofline_hot_path"):
        agg = out[blk]["bytes_per_launch"] / (out["ms_per_step"] * world * 1e-3) / 1e9
        out[blk]["aggregate"] = {"achieved": agg, "frac": agg / HBM_PEAK_GBS, "unit": "GB/s",
                                 "note": "bytes_per_launch x launches / timed seconds (launches of %d contexts overlap)" % a.gpu_streams}
    cache = os.path.join(tempfile.gettempdir(), "tcmi_cpu_baseline_%d_%d.json" % (a.reads, a.level))
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(paths, L, a.mincov, orfs, n_threads=cores)
        try:
            json.dump(out["cpu_baseline"], open(cache, "w"))
        except OSError:
            pass
    elif world > 1 and os.path.exists(cache):                    # (measured by the N = 1 run on this box: the host cores are the same)
        try:
            out["cpu_baseline"] = dict(json.load(open(cache)), cached_from="the N = 1 run on this host")
        except Exception:
            pass
    # the reference's stage B on this box's cores (19.1 s per BuildConsensus call, two calls per run, on the survey
    # container): oracle/tc_oracle.py's build_consensus — the literal per-position walk with the O(L^2) CorrectGFF — as timed by the
    # FASTA check above on the configs[1] matrices (several files at once, one core each)
    if "cpu_baseline" in out and chain_t.get("build_consensus_seconds"):
        bc = chain_t["build_consensus_seconds"]
        one = float(np.median(bc))
        tl = out["cpu_baseline"].get("python_loop_tally_seconds_per_bam_extrapolated")
        out["cpu_baseline"]["python_stage_b"] = {
            "build_consensus_seconds_per_call": one, "calls_per_run": 2, "seconds_per_bam": 2 * one, "files_timed": len(bc),
            "what": "oracle/tc_oracle.py build_consensus (Sequences.py:168-322 + ORFs.py:111-192 restated with the reference's loop structure) on "
                    "this run's configs[1] count matrices, %d worker processes side by side" % min(8, cores)}
        if tl:
            out["cpu_baseline"]["python_reference_standin_seconds_per_bam"] = tl + 2 * one
    # ---- what matters, in short top-level keys -------------------------------------------------------------------------------------
    # `value` is the contract's rate — the inputs (the BAM files' compressed bytes) resident in HBM when the clock starts; the
    # PCIe-inclusive rate from files in the page cache (SURVEY 8-d's wall) stands next to it under a name of its own, never as `value`
    out["value_resident"] = out["value"]
    if "file_to_fasta" in out:
        out["value_file_to_fasta"] = out["file_to_fasta"]["value"]
    if "roofline_tally" in out:
        out["tally_frac_of_hbm_peak_file_path"] = {k: out["roofline_tally"][k]["frac"] for k in ("device_bytes", "counter_bytes", "survey_8d_bytes")}
    for k, leg in (("value_hard_bam", "hard_bam"), ("value_real_bam", "real_bam")):
        if leg in out and "pipelined" in out[leg]:
            out[k] = out[leg]["pipelined"]["value"]
    if "configs0" in out and "error" not in out["configs0"]:
        out["configs0_outputs_equal"] = out["configs0"]["all_four_outputs_equal"]
    if "configs2" in out:
        out["value_configs2"] = out["configs2"]["value"]
        out["configs2_fasta_exact"] = out["configs2"]["fasta_all_timed"]["all_equal_the_oracle_chain"]
    out["kernel_sum_single_stream_us"] = sum(v["us_per_bam"] for k, v in out["cold_kernels"].items() if k not in ("inflate", "hot"))
    return out


def _oracle_chain_worker(job):
    """(a child process: no GPU, no torch) the whole oracle chain on one bench file -> (FASTA text, seconds of the two stage-B passes)."""
    path, orfs, L, mincov, name = job
    import numpy as np
    from types import SimpleNamespace
    from oracle import c_oracle
    reads = c_oracle.read_bam(path)
    t = {}
    text = oracle_fasta_text(SimpleNamespace(mincov=mincov), np, reads, orfs, L, name, timing=t)
    return text, t.get("build_consensus", 0.0), t.get("list_inserts", 0.0)


def oracle_chain_many(jobs, workers):
    """The Python oracle chain (oracle/tc_oracle.py: the reference's own loop structure, ~10 s per file) on several files at once,
    in spawned worker processes (a fork of a process that holds a HIP runtime is not safe)."""
    import multiprocessing as mp
    if len(jobs) <= 1 or workers <= 1:
        return [_oracle_chain_worker(j) for j in jobs]
    # (the workers are CPU only: under rocprofv3 they must not inherit the profiler's preloaded library — a counter pass has hung on their exit)
    hidden = {k: os.environ.pop(k) for k in list(os.environ) if k == "LD_PRELOAD" or k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_"))}
    try:
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            return pool.map(_oracle_chain_worker, jobs)
    finally:
        os.environ.update(hidden)


def check_all_fastas(a, np, paths, fastas, L, mincov, orfs, workers=8, timing=None):
    """EVERY FASTA text of the timed region (K x repeats of them; the i-th is of bench file i mod F) against the ORACLE chain on that
    file as oracle/bam_oracle.c reads it: scalar C tally (oracle/tally_oracle.c), then oracle/tc_oracle.py's list_inserts (the
    pile-up emulator's tokens on the candidate columns) and build_consensus — the Python restatement of the reference's own
    functions, pinned by the golden vectors — for every distinct file (no product code in the checker)."""
    res = oracle_chain_many([(p, [dict(o) for o in orfs], L, mincov, "S%d" % k) for k, p in enumerate(paths)], workers)
    want = [r[0] for r in res]
    if timing is not None:
        timing["build_consensus_seconds"] = [r[1] for r in res]
        timing["list_inserts_seconds"] = [r[2] for r in res]
    bad = [i for i, t in enumerate(fastas) if t != want[i % len(paths)]]
    return {"fastas": len(fastas), "files": len(paths), "all_equal_the_oracle_chain": not bad, "first_mismatch": bad[0] if bad else None,
            "chain": "oracle/bam_oracle.c + tally_oracle.c + tc_oracle.py (list_inserts, build_consensus) on every distinct file",
            "sha256_per_file": [hashlib.sha256(w.encode()).hexdigest()[:16] for w in want]}


def cli_batch_leg(a, paths, ref, orfs, L, sy, fastas):
    """TrueConsense.main(--batch MANIFEST ...) as a user would call it (TrueConsense.py:212-264 per sample): the bench files, cycled,
    FASTA + VCF + corrected GFF + coverage TSV per sample, all written by the native runner's walker threads (csrc/pipeline.cpp).
    One call = argument parsing, GFF / reference reading, runner set-up, the samples, teardown."""
    import subprocess
    tmp = os.path.dirname(paths[0])
    d = os.path.join(tmp, "cli_batch")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "ref.fa"), "w") as fh:
        fh.write(">MN908947.3 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
    head, body = sy.gff_text(orfs)
    with open(os.path.join(d, "f.gff"), "w") as fh:
        fh.write(head + body)
    code = "import sys; sys.path.insert(0, %r); from trueconsense_amd import TrueConsense as c; c.main(sys.argv[1:])" % os.path.dirname(os.path.abspath(__file__))

    def call(n):
        with open(os.path.join(d, "manifest.tsv"), "w") as fh:
            for i in range(n):
                fh.write("\t".join([paths[i % len(paths)], "S%d" % (i % len(paths))] + [os.path.join(d, "o%d.%s" % (i, e)) for e in ("fa", "vcf", "gff", "tsv")]) + "\n")
        argv = ["--batch", os.path.join(d, "manifest.tsv"), "-ref", os.path.join(d, "ref.fa"), "-gff", os.path.join(d, "f.gff"), "-cov", str(a.mincov),
                "--stats", os.path.join(d, "stats.json")]
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", code] + argv, capture_output=True, text=True)     # a process of its own, as from a shell
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout)[-400:])
        return wall, json.load(open(os.path.join(d, "stats.json")))

    # Two manifests, 16 and 80 rounds of the bench files: the difference is what a sample costs once the runner is up (contexts, device
    # arenas and code objects are set up once per call: 0.2 - 0.3 s, varying by tens of ms from call to call — hence manifests that differ
    # by hundreds of samples, and the faster of two calls each).
    n1, n2 = 16 * len(paths), 80 * len(paths)
    try:
        call(2 * len(paths))                                     # (the box's first import of the HIP runtime in a fresh process)
        w1, s1 = min((call(n1) for _ in range(2)), key=lambda x: x[1]["seconds"]["batch"])
        w2, s2 = min((call(n2) for _ in range(2)), key=lambda x: x[1]["seconds"]["batch"])
    except RuntimeError as e:
        return {"error": str(e)}
    same = all(open(os.path.join(d, "o%d.fa" % i)).read() == fastas[i % len(paths)] for i in range(n2))
    sizes = {e: os.path.getsize(os.path.join(d, "o0." + e)) for e in ("fa", "vcf", "gff", "tsv")}
    t1, t2 = s1["seconds"]["batch"], s2["seconds"]["batch"]
    return {"outputs_per_sample": 4, "ms_per_bam": 1e3 * (t2 - t1) / (n2 - n1), "samples": [n1, n2], "seconds_in_runner": [t1, t2],
            "ms_per_bam_with_setup": 1e3 * t2 / n2, "process_wall_seconds": [w1, w2], "stage_busy_seconds": s2["stage_busy_seconds"],
            "decoded_on": s2["decoded_on"], "fasta_files_equal_the_headline_texts": same, "output_bytes_sample0": sizes,
            "note": "python -c 'TrueConsense.main(--batch ...)' in a process of its own; ms_per_bam = (runner seconds of the long manifest - of the short "
                    "one) / (samples more): reader, GPU and walker stages overlapped, the walkers also write VCF, corrected GFF and coverage TSV"}


def configs0_leg(np, sy, ref, orfs, L, tmp, mincov=30, n_reads=10_000, seed=77):
    """BASELINE configs[0] ("10k synthetic 150 bp reads, reference CPU path only (plumbing)"): ONE 10 k-read BAM file through the Python
    stand-in for the reference's whole run (TrueConsense.py:212-264) — oracle/tc_oracle.py end to end with the reference's own loop
    structure: the per-token tally loop (indexing.py:102-132), ListInserts with the region pile-ups (Events.py:5-82), both BuildConsensus
    walks (Sequences.py:168-322, ORFs.py), the writers (Outputs.py, Coverage.py) — timed, beside the product's command line on the same
    file in a process of its own, timed, and the four outputs compared: FASTA text, VCF record lines, the corrected ORF coordinates of
    the GFF, the coverage TSV.  The only leg where a whole stand-in run is timed rather than extrapolated."""
    import subprocess
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    from trueconsense_amd.io import bamwriter
    d = os.path.join(tmp, "configs0")
    os.makedirs(d, exist_ok=True)
    reads = sy.make_reads(ref, n_reads, seed=seed, indel_sites=sy.default_indel_sites(orfs))
    bam = os.path.join(d, "in.bam")
    bamwriter.write_bam(bam, reads, "MN908947.3", L, level=6)
    with open(os.path.join(d, "ref.fa"), "w") as fh:
        fh.write(">MN908947.3 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
    head, body = sy.gff_text(orfs)
    with open(os.path.join(d, "f.gff"), "w") as fh:
        fh.write(head + body)
    # ---- the stand-in, one core: file -> reads (oracle/bam_oracle.c: what pysam's C code does for the reference) -> everything else in Python
    t = {}
    t0 = time.perf_counter()
    rd = c_oracle.read_bam(bam)
    t["read_bam_c"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    counts = np.asarray(orc.tally_matrix(rd, L), np.int64)
    t["tally_python"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    has, ins = orc.list_inserts(counts, mincov, lambda pos1: orc.region_tokens(rd, pos1))
    t["list_inserts"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    rows = [dict(o) for o in orfs]
    cons, cur = orc.build_consensus(mincov, counts, rows, True, ins if has else None, True)
    cons_noins, _ = orc.build_consensus(mincov, counts, [dict(o) for o in orfs], True, ins if has else None, False)
    t["build_consensus_x2"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    want = {"fa": orc.fasta_text("S", mincov, cons), "tsv": orc.coverage_tsv(counts),
            "vcf": orc.vcf_records("MN908947.3", ref, cons_noins, counts, mincov, ins if has else None),
            "gff": [[int(o["start"]), int(o["end"])] for o in (cur.values() if isinstance(cur, dict) else cur)]}
    t["writers"] = time.perf_counter() - t1
    standin = time.perf_counter() - t0
    # ---- the product's command line on the same file (a process of its own, as from a shell: interpreter start and HIP set-up included)
    outs = {e: os.path.join(d, "out." + e) for e in ("fa", "vcf", "gff", "tsv")}
    code = "import sys; sys.path.insert(0, %r); from trueconsense_amd import TrueConsense as c; c.main(sys.argv[1:])" % ROOT
    argv = ["-i", bam, "-ref", os.path.join(d, "ref.fa"), "-gff", os.path.join(d, "f.gff"), "-cov", str(mincov), "-name", "S", "-o", outs["fa"],
            "-vcf", outs["vcf"], "-ogff", outs["gff"], "-doc", outs["tsv"], "--stats", os.path.join(d, "stats.json")]
    walls = []
    for _ in range(2):                                               # (the second call: code objects and page cache warm)
        t1 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", code] + argv, capture_output=True, text=True)
        walls.append(time.perf_counter() - t1)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
    stats = json.load(open(os.path.join(d, "stats.json")))
    got = {"fa": open(outs["fa"]).read(), "tsv": open(outs["tsv"]).read(),
           "vcf": "".join(ln + "\n" for ln in open(outs["vcf"]).read().split("\n") if ln and not ln.startswith("#")),
           "gff": [[int(f[3]), int(f[4])] for f in (ln.split("\t") for ln in open(outs["gff"]).read().split("\n") if ln and not ln.startswith("#"))]}
    same = {k: bool(got[k] == want[k]) for k in want}
    return {"reads": n_reads, "positions": L, "indel_sites": len(sy.default_indel_sites(orfs)), "accepted_inserts": len(ins) if has else 0,
            "python_standin_seconds": standin, "python_standin_stage_seconds": t, "cores": 1,
            "product_cli_process_seconds": min(walls), "product_cli_process_seconds_runs": walls, "product_cli_in_process_seconds": stats.get("seconds"),
            "outputs_equal": same, "all_four_outputs_equal": all(same.values()), "vcf_records": want["vcf"].count("\n"),
            "note": "configs[0]: one 10k-read BAM (indel carriers at CDS boundaries included); oracle/tc_oracle.py end to end (the reference's loop structure, "
                    "single core; reading the BAM is C on both sides) beside `TrueConsense -i .. -o .. -vcf .. -ogff .. -doc ..` in a process of its own"}


def configs2_leg(a, np, sy, runner, ctx, ref, orfs, L, tmp, rank, workers, deferred=None):
    """BASELINE configs[2] (1M reads, 2 % of them carrying an insertion or deletion at CDS starts / ends): two such files, their compressed
    bytes resident in HBM, a queue of them through the headline's runner; EVERY FASTA against the whole Python oracle chain (inserts
    included) on its file."""
    from trueconsense_amd.engine import DeviceBam
    from trueconsense_amd.io import bamwriter
    sites = sy.default_indel_sites(orfs)
    t0 = time.perf_counter()
    paths = []
    for k in range(2):
        reads = sy.make_reads(ref, a.reads, seed=3000 + 1000 * rank + k, indel_sites=sites)
        p = os.path.join(tmp, "c2_%d.bam" % k)
        bamwriter.write_bam(p, reads, "MN908947.3", L, level=a.level)
        paths.append(p)
        del reads
    t_gen = time.perf_counter() - t0
    db = [DeviceBam(p).to_device(ctx) for p in paths]
    runner.run_resident([db[i % 2] for i in range(16)], ref_len=L)
    n = 384                                                          # (0.2 s of work: 64 files — 35 ms, mostly the pipeline's fill and drain — measured 47 - 61 M from run to run)
    ctx.sync()
    t1 = time.perf_counter()
    texts = runner.run_resident([db[i % 2] for i in range(n)], names=["S%d" % (i % 2) for i in range(n)], ref_len=L)
    ctx.sync()
    dt = time.perf_counter() - t1
    for d in db:
        d.close()
    res = {}

    def check():                                                     # (the Python oracle chain on both files: after the GPU legs)
        res["fasta_all_timed"] = check_all_fastas(a, np, paths, texts, L, a.mincov, orfs, workers=workers)
    if deferred is None:
        check()
    else:
        deferred.append(check)
    res.update({"value": L * n / dt, "unit": "positions/s", "ms_per_bam": 1e3 * dt / n, "bams": n, "files": 2,
            "consensus_len": [len(t.split("\n")[1]) for t in texts[:2]], "decoded_on": dict(runner.decoded_on),
            "input_generation_seconds_outside_clock": t_gen,
            "note": "configs[2]: 1M reads per BAM, 1 %% insertion + 1 %% deletion carriers at CDS boundaries; %d files' compressed bytes resident in HBM, "
                    "%d of them overlapped through the headline's runner; the insert candidates' tokens are voted on from entries the device "
                    "collects in the resident inflated stream (ins_entries_kernel)" % (2, n)})
    return res


def hard_bam_leg(a, single, sctx, _ffi, np, sy, ref, L, tmp, runner, ctx, kind="hard"):
    """tools/hard_bam.py's file shape at the bench's size: Illumina-style names (all distinct), qualities drawn from four bins —
    it compresses ~6 : 1 instead of 35 : 1, so a BGZF block holds six times the symbols and a fifth of its matches reach back
    further than the decoder's LDS ring.  One BAM at a time through the same runner; kernel times from HIP events.
    kind "real": qualities drawn like an Illumina run's (a peak at Q36, a tail down to Q2) — 2.5 : 1, what real files compress like:
    21 000 tokens and 26 KB of payload per block, two deflate streams in most blocks."""
    from trueconsense_amd.engine import DeviceBam, Walker
    from trueconsense_amd.io import bamwriter
    from oracle import c_oracle
    orfs = sy.make_reference()[1]
    single_walker = Walker([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    n = a.reads
    reads = sy.make_reads(ref, n, seed=4242)
    qual, names = like_real_data(np, kind, n, seed=1)
    path = os.path.join(tmp, kind + ".bam")
    t0 = time.perf_counter()
    bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=a.level, qual=qual, names=names)
    t_gen = time.perf_counter() - t0
    d = DeviceBam(path)
    fb, ib, nb = d.file_bytes, d.inflated_bytes, d.n_blocks
    d.close()
    single.run([path], ref_len=L)
    lat = []
    for _ in range(3 if kind == "hard" else 2):
        t1 = time.perf_counter()
        text = single.run([path], names=["H"], ref_len=L)[0]
        lat.append(time.perf_counter() - t1)
    sctx.profile(False)
    sctx.profile(True)
    n_k = 4
    single.run([path] * n_k, ref_len=L)
    k = kernel_times([sctx], _ffi, n_k)
    sctx.profile(False)
    # ... and overlapped like the headline: the file's bytes resident in HBM, a queue of 48 of it through the headline's runner
    dres = DeviceBam(path).to_device(ctx)
    runner.run_resident([dres] * 8, ref_len=L)
    ctx.sync()
    t1 = time.perf_counter()
    texts = runner.run_resident([dres] * 48, names=["H"] * 48, ref_len=L)
    ctx.sync()
    piped = (time.perf_counter() - t1) / 48
    piped_same = all(t == text for t in texts)
    dres.close()
    counts = c_oracle.tally(reads, L)                           # (the checker, outside every clock: scalar C tally + call, host walk)
    plain, alt, flags = c_oracle.call(counts, a.mincov, True)
    want = ">H mincov=%d\n%s\n" % (a.mincov, single_walker(plain[:L], alt[:L], flags[:L])[0].decode("ascii"))
    return {"reads": n, "fasta_bit_exact": bool(text == want), "file_bytes": fb, "inflated_bytes": ib, "ratio": ib / max(1, fb), "bgzf_blocks": nb,
            "seconds_per_bam": min(lat), "value": L / min(lat), "unit": "positions/s (one BAM at a time, file -> FASTA)",
            "pipelined": {"value": L / piped, "unit": "positions/s", "ms_per_bam": 1e3 * piped, "fastas_equal": bool(piped_same),
                          "note": "48 x the same file, compressed bytes resident in HBM, through the headline's runner"},
            "inflate_us": k["inflate"]["us_per_bam"], "kernels_us": {x: round(v["us_per_bam"], 1) for x, v in k.items()},
            "fasta_sha256": hashlib.sha256(text.encode()).hexdigest()[:16], "input_generation_seconds_outside_clock": t_gen}


def cold_rooflines(a, timed, cold, file_bytes, inflated_bytes, reads0):
    """Per-kernel bytes against the 8 TB/s HBM roofline for the kernels of the file -> FASTA path.  `avg_launch_us` is the HIP-event
    time of the launches of the TIMED region (three contexts: kernels of different BAMs share the GPU, which stretches each
    of them — rocprofv3's per-kernel average of the same command, profiles/, shows the same stretch); `single_stream_us` is the
    same kernel with nothing else on the GPU.  `roofline` is the kernel pair with the largest share of the GPU time (BGZF inflate:
    bound by instruction issue / latency, not by HBM — the fraction says how far from HBM-bound it is); `roofline_hot_path` is the
    tally path proper (the packer, which reads the BAM-native bytes SURVEY 8-d counts).  `traffic` = HBM bytes from
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this round (profiles/traffic.json), not measured in this run."""
    import numpy as np
    n = int(reads0["n_reads"])
    alg_reads = int(np.sum(12 + 4 * np.diff(reads0["cigar_off"].astype(np.int64)) + (reads0["l_qseq"].astype(np.int64) + 1) // 2))
    traffic = {}
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp))
        except Exception:
            traffic = {}

    r3_key = next((k for k in ("round6", "round5", "round4", "round3") if k in traffic), None)
    r3 = traffic.get(r3_key, {})
    r3k = r3.get("kernels", {})

    def pmc_bytes(names):
        return sum(r3k[x]["hbm_bytes"] for x in names) if all(x in r3k for x in names) else None

    def block(kernel, key, bytes_moved, note, alg=None, pmc_of=()):
        us, us1 = timed[key]["us_per_bam"], cold[key]["us_per_bam"]
        ach = bytes_moved / (us * 1e-6) / 1e9 if us > 0 else 0.0
        tr = pmc_bytes(pmc_of)
        b = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "single_stream_us": us1, "single_stream_frac": (bytes_moved / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS) if us1 > 0 else None,
             "traffic": tr, "traffic_source": ("profiles/traffic.json %s: FETCH_SIZE + WRITE_SIZE of " % r3_key + " + ".join(pmc_of) + ", " + r3.get("source", "")) if tr else None,
             "bytes_per_launch": bytes_moved, "avg_launch_us": us, "note": note}
        if alg is not None:
            b["algorithmic"] = {"bytes_per_launch": alg, "achieved": alg / (us * 1e-6) / 1e9 if us > 0 else 0.0,
                                "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else 0.0}
        return b

    inflated = inflated_bytes or 273 * n
    # SURVEY 8-a1 / a2 (BuildIndex, parse_query_sequences) are ALL the kernels behind the decode: pk_index (record index + classify),
    # pk_place (prefix + planes), pk_pack (chunks) and the tally
    for key in ("hot",):
        timed[key] = {"us_per_bam": sum(timed[k]["us_per_bam"] for k in ("records", "pack_classify", "pack", "tally"))}
        cold[key] = {"us_per_bam": sum(cold[k]["us_per_bam"] for k in ("records", "pack_classify", "pack", "tally"))}
    hot = block("pk_index + pk_place + pk_pack + tally_planes_kernel (the several-kernel path: rec_* + pk_classify + pk_scan + pk_scatter + pk_pack + pk_planes + tally)",
                "hot", alg_reads + 2 * 52 * n + 28 * 29903,
                "SURVEY 8-d algorithmic bytes of the reads (12 + 4 n_cigar + l/2 each) + the 52 B per read written by the packer and read by "
                "the tally + the count matrix; the bases are read out of the inflated BAM stream (289 B per record: names and qualities ride "
                "along in the 64-byte sectors)", alg=alg_reads + 28 * 29903, pmc_of=("pk_index", "pk_place", "pk_pack", "tally_planes_kernel"))
    out = {"roofline": block("bgzf_symbols + bgzf_copy", "inflate", file_bytes + inflated,
                             "compressed bytes read + inflated bytes written per BAM (the tokens between the two kernels — 4 bytes per literal / match — "
                             "are traffic, not algorithmic bytes: the speculating lanes park theirs in scattered 4-byte stores, which is most of what the "
                             "counters see beyond the stream); Huffman symbols decoded 32 lanes per block speculatively (latency-bound), LZ77 copies through "
                             "an LDS ring (issue-bound): see `issue`; neither is an HBM kernel", pmc_of=("bgzf_symbols", "bgzf_copy")),
           "roofline_hot_path": hot}
    # What does bound these kernels.  The chip's issue rate (MI355X_MICROARCH.md: four SIMD-32 per CU, a wave64 VALU instruction takes
    # a SIMD two cycles -> TWO vector wave-instructions per CU and cycle; one scalar unit per CU -> ONE scalar wave-instruction per CU
    # and cycle) gives every kernel an issue floor: max(VALU / 2, SALU) / (CUs x clock).  Where the wave-cycles go instead, from the
    # committed counter passes: `parked` = 1 - (SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY) / SQ_WAVE_CYCLES — waves that neither issue nor
    # wait for their instruction's unit: at an s_waitcnt or a barrier, i.e. on memory.
    try:
        import torch
        pr = torch.cuda.get_device_properties(0)
        n_cu, ghz = int(pr.multi_processor_count), float(getattr(pr, "clock_rate", 2400000)) / 1e6
    except Exception:
        n_cu, ghz = 256, 2.4
    issue = {"compute_units": n_cu, "clock_ghz": ghz, "source": r3.get("source"),
             "rates": "2 VALU + 1 SALU wave-instructions per CU and cycle (4 SIMD-32, a wave64 VALU instruction = 2 cycles of one SIMD)", "kernels": {}}
    cold_key = {"bgzf_symbols": "inflate_symbols", "bgzf_copy": "inflate_copy", "pk_index": "pack_classify", "tally_planes_kernel": "tally", "call_kernel": "call"}
    floor_sum = 0.0
    for name, c in r3k.items():
        wi = c.get("wave_insts")
        if not wi:
            continue
        valu_us = wi.get("valu", 0) / (2.0 * n_cu * ghz * 1e3)
        salu_us = (wi.get("salu", 0) + wi.get("smem", 0)) / (n_cu * ghz * 1e3)
        e = {"wave_insts_per_launch": c.get("wave_insts_total"), "mix": wi, "valu_floor_us": valu_us, "salu_floor_us": salu_us,
             "issue_floor_us": max(valu_us, salu_us), "parked": c.get("parked")}
        floor_sum += e["issue_floor_us"]
        us1 = cold.get(cold_key.get(name, ""), {}).get("us_per_bam", 0)
        if us1 > 0:
            e.update(kernel_us=us1, floor_over_kernel_time=e["issue_floor_us"] / us1)
        issue["kernels"][name] = e
    if r3.get("wave_insts_per_bam"):
        tv = sum(c.get("wave_insts", {}).get("valu", 0) for c in r3k.values())
        ts = sum(c.get("wave_insts", {}).get("salu", 0) + c.get("wave_insts", {}).get("smem", 0) for c in r3k.values())
        issue["all_kernels_of_a_bam"] = {"wave_insts": r3["wave_insts_per_bam"], "valu": tv, "salu": ts,
                                         "valu_floor_ms": tv / (2.0 * n_cu * ghz * 1e6), "salu_floor_ms": ts / (n_cu * ghz * 1e6),
                                         "issue_floor_ms": floor_sum / 1e3, "ms_per_step_pipelined": None,
                                         "hbm_bytes_per_bam": r3.get("hbm_bytes_per_bam")}
    if issue["kernels"]:
        out["roofline"]["issue"] = issue
    return out


def oracle_fasta_text(a, np, reads, orfs, L, name, timing=None):
    """The oracle chain on one file's reads -> its FASTA text: scalar C tally (oracle/tally_oracle.c), reference-pinned list_inserts
    (with the pileup emulator's tokens on the candidate columns) and build_consensus (oracle/tc_oracle.py)."""
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    counts = c_oracle.tally(reads, L)
    if "sorted_max_span" in reads:
        span = int(reads["sorted_max_span"])
    else:                                                        # (reads decoded from a file by oracle/bam_oracle.c: the longest reference span)
        cg = np.asarray(reads["cigar"], np.int64)
        cs = np.concatenate(([0], np.cumsum((cg >> 4) * np.isin(cg & 15, (0, 2, 3, 7, 8)))))
        co = np.asarray(reads["cigar_off"], np.int64)
        span = int((cs[co[1:]] - cs[co[:-1]]).max()) if len(co) > 1 else 1
    keep = np.flatnonzero((np.asarray(reads["flag"]) & 4) == 0) if "sorted_max_span" not in reads else None
    pos_sorted = np.asarray(reads["pos"]) if keep is None else np.asarray(reads["pos"])[keep]

    def tokens_at(pos1):
        # (the emulator loops over reads in Python: hand it only the sorted reads that can reach the column)
        c = pos1 - 1
        i0, i1 = int(np.searchsorted(pos_sorted, c - span + 1, "left")), int(np.searchsorted(pos_sorted, c, "right"))
        if keep is not None:                                     # (mapped reads come first in a sorted BAM: indices into the file's order)
            i0, i1 = (int(keep[i0]) if i0 < len(keep) else len(reads["pos"])), (int(keep[i1 - 1]) + 1 if i1 > 0 else 0)
        co, so, qo = (np.asarray(reads[k]) for k in ("cigar_off", "seq_off", "qual_off"))
        sub = {"n_reads": i1 - i0, "pos": reads["pos"][i0:i1], "flag": reads["flag"][i0:i1], "l_qseq": reads["l_qseq"][i0:i1],
               "tid": reads["tid"][i0:i1], "cigar_off": (co[i0:i1 + 1] - co[i0]).astype(np.uint64), "cigar": reads["cigar"][int(co[i0]):int(co[i1])],
               "seq_off": (so[i0:i1 + 1] - so[i0]).astype(np.uint64), "seq": reads["seq"][int(so[i0]):int(so[i1])],
               "qual_off": (qo[i0:i1 + 1] - qo[i0]).astype(np.uint64), "qual": reads["qual"][int(qo[i0]):int(qo[i1])]}
        return orc.region_tokens(sub, pos1)

    t0 = time.perf_counter()
    has, ins = orc.list_inserts(counts, a.mincov, tokens_at)
    t1 = time.perf_counter()
    want, _ = orc.build_consensus(a.mincov, counts.astype(np.int64), [dict(o) for o in orfs], True, ins if has else None, True)
    if timing is not None:
        timing["list_inserts"] = t1 - t0
        timing["build_consensus"] = time.perf_counter() - t1
    return orc.fasta_text(name, a.mincov, want)


def check_fasta(a, np, fasta_text, reads0, ref, orfs, L):
    """FASTA text of bench file 0 from the timed path vs the oracle chain on the reads the file was written from."""
    want_text = oracle_fasta_text(a, np, reads0, orfs, L, "S0")
    return {"fasta_bit_exact": bool(fasta_text == want_text), "fasta_sha256": hashlib.sha256(fasta_text.encode()).hexdigest()[:16],
            "consensus_len": len(fasta_text.split("\n")[1])}


def resident_leg(a, ctx0, Pipeline, local_rank, np, _ffi, sy, ref, orfs, L, rank, fence):
    """Reads of `--resident-batch` BAMs already packed in HBM as ONE read set (BAM b at positions shifted by
    b * 29 952): per step one tally launch (whose first blocks call the previous step's matrix) + host walks, through
    the native pipeline; repeated until at least one second has been timed.  HIP events around every
    `--profile-every`-th launch give the kernel time for the roofline block."""
    from concurrent.futures import ThreadPoolExecutor
    B = max(1, a.resident_batch)
    pos_stride = (L + 255) // 256 * 256
    n_walkers = max(1, min(16, (os.cpu_count() or 1) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    pipe = Pipeline(local_rank, slots=a.slots, walkers=n_walkers)
    ctxs = [pipe.slot_context(k) for k in range(a.slots)]
    for kv in a.ctx_option:
        k, v = kv.split("=")
        for c in ctxs:
            c.set_option(k, int(v))
    sites = sy.default_indel_sites(orfs) if a.indels else None
    readsets, hreads = [], []
    with ThreadPoolExecutor(min(8, B)) as ex:
        for b in range(2):
            group = list(ex.map(lambda k: sy.make_reads(ref, a.reads, seed=5000 + 1000 * rank + b * B + k, indel_sites=sites), range(B)))
            readsets.append(pipe.ctx.upload(group[0]) if B == 1 else pipe.ctx.upload_batch(group, pos_stride))
            hreads.extend(group if a.indels else [None] * B)
            del group
    pipe.set_orfs([o["start"] for o in orfs], [o["end"] for o in orfs], [1] * len(orfs))
    n = a.resident_steps
    stride = L + 1 + (4096 if a.indels else 64)
    buf = np.empty(n * B * stride, np.uint8)
    buf.fill(0)
    hr = [hreads[(i % 2) * B + k] for i in range(n) for k in range(B)] if a.indels else None

    def run():
        pipe.run([readsets[i % 2] for i in range(n)], L, a.mincov, True, host_reads=hr, extra=stride - L - 1, batch=B,
                 pos_stride=pos_stride, out=buf)

    run()                                                        # warm
    for c in ctxs:
        c.set_option("profile_every", a.profile_every)
        c.profile(True)
    fence()
    t0 = time.perf_counter()
    repeats = 0
    while repeats < 1 or time.perf_counter() - t0 < 1.0:
        run()
        repeats += 1
    fence()
    dt = time.perf_counter() - t0
    tally_ms = tally_n = 0
    for c in ctxs:
        m, k = c.profile_get(_ffi.K_TALLY)
        tally_ms += m
        tally_n += k
        c.profile(False)
    tally_us = 1e3 * tally_ms / max(1, tally_n)
    real = readsets[0].device_bytes + 28 * L * B                 # what one launch reads (packed reads) + one write of the matrices
    alg = readsets[0].algorithmic_bytes + 28 * L * B             # the BAM-form bytes (91 B per 150M read)
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            rk = next((k for k in ("round6", "round5", "round4") if k in tj), None)
            per = ((tj.get(rk) or {}).get("kernels", {}).get("tally_planes_kernel") or {}).get("hbm_bytes")
            traffic = per * B if per else None                      # (a round-1 figure stood here until round 3: dropped rather than quoted stale)
            traffic_src = ("profiles/traffic.json %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tally_planes_kernel per 1M-read BAM (one BAM per launch) x batch" % rk) if per else None
        except Exception:
            traffic = None
    achieved = real / (tally_us * 1e-6) / 1e9 if tally_us > 0 else 0.0
    res = {"resident": {"value": L * n * repeats * B / dt, "unit": "positions/s", "ms_per_step": 1e3 * dt / (n * repeats),
                        "steps": n, "repeats": repeats, "bams_per_step": B, "seconds": dt,
                        "note": "reads already packed in HBM; pack + H2D + decode are NOT in this figure"},
           "roofline": {"kernel": "tally_planes_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "bytes_per_launch": real, "bytes_note": "device bytes of the packed read set (read once) + 28 B x positions (count matrix)",
                        "avg_launch_us": tally_us, "launches_timed": tally_n,
                        "algorithmic": {"bytes_per_launch": alg, "achieved": alg / (tally_us * 1e-6) / 1e9 if tally_us > 0 else 0.0,
                                        "frac": (alg / (tally_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tally_us > 0 else 0.0,
                                        "note": "SURVEY 8-d bytes (12 + 4 n_cigar + l/2 per read): the kernel reads the packed 2-bit planes, fewer bytes than this"}}}
    for r in readsets:
        r.free()
    pipe.close()
    return res


if __name__ == "__main__":
    main()
