"""The N > 1 paths: shard bookkeeping, and the one-BAM-split all-reduce over a real process
group (gloo, world_size 2, 127.0.0.1).  On CPU the per-shard tally is the oracle's; the gpu-marked
test runs the HIP tally in both ranks on one MI355X and still reduces through gloo (RCCL needs one
GPU per rank; the driver's 8-GPU run covers that)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, use_gpu, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from oracle import c_oracle
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _work(rank, world, use_gpu, q, td, sy, c_oracle)
    except Exception as e:                                   # make a failing rank visible to the parent
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


def _consensus_case():
    """An indel-carrier read set whose insert-candidate columns lie around the middle of the file (where two ranks' ranges meet):
    accepted insertions of 1, 2 and 14 bases (the last too long for an entry's key: its bases travel as text), a rejected one."""
    from trueconsense_amd import synthetic as sy
    ref, orfs = sy.make_reference(L=4000, cds=[(100, 1900), (2100, 3900)])
    sites = [(1900, "I", "A", 0.9), (1990, "I", "GT", 0.7), (2005, "I", "ACGTACGTACGTAC", 0.8), (2010, "D", 3, 0.6), (2100, "I", "C", 0.4), (3000, "I", "TT", 0.95)]
    reads = sy.make_reads(ref, 6000, seed=91, indel_sites=sites)
    return ref, orfs, reads


def _oracle_fasta(reads, orfs, L, mincov):
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    counts = c_oracle.tally(reads, L)
    has, ins = orc.list_inserts(counts, mincov, lambda pos1: orc.region_tokens(reads, pos1))
    cons, _ = orc.build_consensus(mincov, counts.astype(np.int64), [dict(o) for o in orfs], True, ins if has else None, True)
    return orc.fasta_text("S", mincov, cons), ins if has else {}


def _work_consensus(rank, world, use_gpu, td, c_oracle):
    """configs[4] to the consensus: two ranks, candidate columns on both sides of the boundary; FASTA = the oracle chain's on the whole file."""
    import torch.distributed as dist
    ref, orfs, reads = _consensus_case()
    L = len(ref)
    rows = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs]
    want, ins = _oracle_fasta(reads, orfs, L, 30)
    assert len(ins) >= 4 and any(len(next(iter(v.values()))) > 12 for v in ins.values())
    if use_gpu:
        import tempfile
        from trueconsense_amd.io import bamwriter
        path = os.path.join(tempfile.gettempdir(), "tcmi_cons_%d.bam" % os.getppid())
        ok = True
        for split in (False, True):
            if rank == 0:
                bamwriter.write_bam(path, reads, "r", L, level=6, block=3000, split_records=split)
            dist.barrier()
            text = td.consensus_split_bamfile(path, L, rows, 30, True, "S", rank, world, device=0)
            ok = ok and (text == want if rank == 0 else text is None)
            dist.barrier()
        if rank == 0:
            os.remove(path)
        return ok
    from tests import entries_py
    shard = td.shard_reads(reads, rank, world)
    for k in ("next_tid", "next_pos", "tlen"):
        if reads.get(k) is not None:
            a, b = td.read_range(int(reads["n_reads"]), rank, world)
            shard[k] = reads[k][a:b]

    def step_fn(what):
        if what == "n_blocks":
            return world
        if what[0] == "call":
            return c_oracle.call(what[1], what[2], what[3])
        return c_oracle.tally(shard, L)
    j0 = td.read_range(int(reads["n_reads"]), rank, world)[0]
    text = td.consensus_split_bamfile("unused", L, rows, 30, True, "S", rank, world, step_fn=step_fn,
                                      entries_fn=lambda pos: entries_py.entries_for(shard, pos, j0=j0))
    ok = text == want if rank == 0 else text is None
    # ONE rank cannot collect its entries (TCMI_E_UNSUPPORTED: a read of more than 512 positions in its range, ONT data): the ranks
    # agree before the gather — nobody is left waiting in it — and rank 0 sweeps the file on the host for the tokens: the same FASTA
    import tempfile
    from trueconsense_amd import _ffi
    from trueconsense_amd.io import bamwriter
    path = os.path.join(tempfile.gettempdir(), "tcmi_cons_cpu_%d.bam" % os.getppid())
    if rank == 0:
        bamwriter.write_bam(path, reads, "r", L, level=1)
    dist.barrier()

    def refusing(code):
        def fn(pos):
            if rank == 1:
                raise _ffi.TcmiError(code, "rank 1 cannot collect its entries")
            return entries_py.entries_for(shard, pos, j0=j0)
        return fn
    text = td.consensus_split_bamfile(path, L, rows, 30, True, "S", rank, world, step_fn=step_fn, entries_fn=refusing(_ffi.E_UNSUPPORTED))
    ok = ok and (text == want if rank == 0 else text is None)
    try:                                                             # any other failure: every rank raises, none hangs
        td.consensus_split_bamfile(path, L, rows, 30, True, "S", rank, world, step_fn=step_fn, entries_fn=refusing(_ffi.E_NOMEM))
        ok = False
    except _ffi.TcmiError as e:
        ok = ok and e.code == _ffi.E_NOMEM
    dist.barrier()
    if rank == 0:
        os.remove(path)
    return ok


def _work(rank, world, use_gpu, q, td, sy, c_oracle):
    ref, orfs = sy.make_reference(L=4000, cds=[(100, 3000)])
    reads = sy.make_reads(ref, 9001, seed=77, indel_sites=[(800, "D", 2, 0.5), (1500, "I", "AC", 0.7)])
    L = len(ref)
    fn = None if use_gpu else (lambda shard, L_: c_oracle.tally(shard, L_))
    got = td.tally_split_bam(reads, L, rank, world, device=0, tally_fn=fn)
    want = c_oracle.tally(reads, L)
    ok = bool(np.array_equal(got, want))
    # the reduce-to-root form of the same exchange (what bench.py --split-bam does per step): only rank 0 gets the sum
    import torch
    part = torch.from_numpy(np.ascontiguousarray(c_oracle.tally(td.shard_reads(reads, rank, world), L).T.astype(np.int32)))
    mine = part.clone()
    td.reduce_counts(part, dst=0)
    if rank == 0:
        ok = ok and bool(np.array_equal(part.numpy().T, want))
    else:
        del mine                                             # (gloo leaves the non-root buffers unspecified)
    if use_gpu:
        # ... and from ONE FILE: every rank decodes only the records that start in its range of the file's BGZF blocks
        import tempfile
        from trueconsense_amd.io import bamwriter
        path = os.path.join(tempfile.gettempdir(), "tcmi_split_%d.bam" % os.getppid())
        for split in (False, True):                          # blocks cut on record boundaries / filled to the brim
            if rank == 0:
                bamwriter.write_bam(path, reads, "r", L, level=6, block=4000, split_records=split)
            import torch.distributed as dist
            dist.barrier()
            both = td.tally_split_bamfile(path, L, rank, world, device=0)
            ok = ok and bool(np.array_equal(both, want))
            dist.barrier()
        # ranges that do not join (here: the rule is made to say so) are TCMI_E_UNSUPPORTED on every rank — what a caller catches to
        # fall back to the host reader —, before anybody enters the collective
        from trueconsense_amd import _ffi
        real = td.check_range_anchors
        td.check_range_anchors = lambda ranges, n: "rank 1's block range starts a record at stream offset 7, the range in front ends its last record at 9"
        try:
            td.tally_split_bamfile(path, L, rank, world, device=0)
            ok = False
        except _ffi.TcmiError as e:
            ok = ok and e.code == _ffi.E_UNSUPPORTED and "host reader" in str(e)
        finally:
            td.check_range_anchors = real
        dist.barrier()
        if rank == 0:
            os.remove(path)
    ok = ok and _work_consensus(rank, world, use_gpu, td, c_oracle)
    q.put((rank, ok, int(got[:, 0].sum())))


def _run(world, use_gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert len({r[2] for r in res}) == 1


def test_shard_bookkeeping():
    from trueconsense_amd import distributed as td
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [td.read_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[k][1] == edges[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1
    for nb in (0, 1, 5, 4187):
        for world in (1, 2, 8):
            r = [td.block_range(nb, k, world) for k in range(world)]
            assert r[0][0] == 0 and sum(c for _, c in r) == nb and all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(world - 1))
    assert td.shard_items(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((td.shard_items(512, r, 8) for r in range(8)), [])) == list(range(512))


def test_shards_partition_the_reads():
    from tests import synth_small as ss
    from trueconsense_amd import distributed as td
    from oracle import c_oracle
    import json
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "outputs.json")))[3]
    reads = ss.reads_from_spec(case["spec"])
    L = len(case["counts"])
    for world in (2, 3, 5):
        parts = [c_oracle.tally(td.shard_reads(reads, r, world), L) for r in range(world)]
        assert np.array_equal(sum(parts), np.array(case["counts"]))


def test_split_bam_allreduce_gloo_world2():
    _run(2, use_gpu=False)


@pytest.mark.gpu
def test_split_bam_allreduce_gpu_tally_gloo_world2():
    _run(2, use_gpu=True)


def _bench_ranks(extra, env_extra=None, nproc=2, launcher=True):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed with two ranks on
    ONE GPU over gloo (TCMI_BENCH_REHEARSE=1: RCCL wants a GPU per rank); returns rank 0's JSON line.
    launcher=False: plain `python bench.py --gpus N` — the script starts its own ranks (bench.spawn_ranks)."""
    import json
    import subprocess
    env = dict(os.environ, TCMI_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] if launcher else [sys.executable]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_many_bam_shard_two_ranks_rehearsal():
    d = _bench_ranks(["--steps", "3", "--warmup", "1", "--reads", "40000", "--files", "2", "--no-cpu-baseline", "--no-resident"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["fasta_bit_exact"] is True
    assert d["value"] > 0 and d["unit"] == "positions/s" and "roofline" in d


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` (plain python, as the driver's N = 1 command is shaped): the flag alone must shard — two child ranks,
    started before the parent touches a GPU, one JSON line with n_gpus 2."""
    d = _bench_ranks(["--steps", "4", "--warmup", "1", "--reads", "40000", "--files", "2", "--no-cpu-baseline", "--no-resident", "--no-hard-bam",
                      "--no-cli-batch", "--no-configs2", "--no-configs0", "--min-seconds", "0"], launcher=False)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["fasta_bit_exact"] is True
    assert d["ms_per_step_per_rank"]["max"] >= d["ms_per_step_per_rank"]["min"] > 0


def test_bench_spawn_ranks_ends_the_others_when_one_fails(tmp_path, monkeypatch):
    """bench.spawn_ranks (CPU: the children are stand-ins): the worst exit code comes back, and a rank that sits in a 'collective' is
    ended once another has failed."""
    import bench
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "mode = sys.argv[1]\n"
                      "if mode == 'ok': sys.exit(0)\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(600)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(bench.time, "monotonic", lambda base=__import__("time").monotonic: base() * 20.0)     # (the 20 s of grace in one)
    monkeypatch.setattr(sys, "argv", ["bench.py", "ok"])
    assert bench.spawn_ranks(3) == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "hang"])
    t0 = __import__("time").perf_counter()
    assert bench.spawn_ranks(3) != 0
    assert __import__("time").perf_counter() - t0 < 30


@pytest.mark.gpu
def test_bench_split_bam_from_one_file_two_ranks_rehearsal():
    d = _bench_ranks(["--split-bam", "--from-file", "--steps", "3", "--warmup", "1", "--reads", "40000"])
    assert d["n_gpus"] == 2 and d["coverage_sum"] == d["coverage_sum_expected"] == 150 * 40000 * 2
    assert d["consensus_len"] == 29903 and d["config"]["blocks_per_rank"][0] > 0 and d["fasta_bit_exact"] is True


@pytest.mark.gpu
def test_bench_split_bam_two_ranks_rehearsal():
    d = _bench_ranks(["--split-bam", "--steps", "3", "--warmup", "1", "--reads", "40000"])
    assert d["n_gpus"] == 2 and d["coverage_sum"] == d["coverage_sum_expected"] == 150 * 40000 * 2
    assert d["consensus_len"] == 29903 and "gloo" in d["config"]["collective"]


@pytest.mark.gpu
def test_bench_many_bam_shard_four_ranks_rehearsal():
    """The driver's N = 8 launch shape at tiny sizes, to shake out collisions of temporary files, ports and thread counts between
    the ranks of one host — with FOUR ranks: the GPU boxes allow at most six processes on a card at once, and the test runner
    itself holds one (six ranks were killed by the box's process guard)."""
    d = _bench_ranks(["--steps", "2", "--warmup", "1", "--reads", "20000", "--files", "2", "--no-cpu-baseline", "--no-resident", "--no-hard-bam",
                      "--no-cli-batch", "--no-configs2", "--min-seconds", "0"], nproc=4)
    assert d["n_gpus"] == 4 and d["steps"] == 2 and d["scaling"] == "weak" and d["fasta_bit_exact"] is True
    assert d["fasta_all_timed"]["all_equal_the_oracle_chain"] is True


@pytest.mark.gpu
def test_configs4_eight_ranks_in_turn_on_one_gpu(tmp_path):
    """BASELINE configs[4]'s shape — ONE file over EIGHT ranks — through the code every rank of the real job runs (tcmi_split_step: range +
    the block behind it, range table, the root's pairwise check of the joins), the ranks played one after the other on the one GPU
    (distributed.split_ranks_in_turn; tools/configs4_full.py runs it at 8 x 6.25 M reads).  (1) 8 x 1 M plain reads, htslib-style
    blocks: counts = the scalar C tally, FASTA = the oracle chain.  (2) indel carriers, records across blocks, THREE ranks: the insert
    candidates' entries gathered in rank order, FASTA = the oracle chain (Events.py:47-82)."""
    from oracle import c_oracle
    from oracle import tc_oracle as orc
    from trueconsense_amd import distributed as td
    from trueconsense_amd import synthetic as sy
    from trueconsense_amd.io import bamwriter
    ref, orfs = sy.make_reference()
    L = len(ref)
    tile = L - 150 + 1
    path = str(tmp_path / "c4.bam")
    want = np.zeros((L, 7), np.int64)
    n, m = 1_000_000, 8
    for k in range(m):
        reads = sy.make_reads(ref, n, seed=9100 + k, start_range=(tile * k // m, tile * (k + 1) // m))
        bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=1, part=(k == 0, k == m - 1), first_id=k * n)
        want += c_oracle.tally(reads, L)
    rows = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs]
    tm = {}
    text, counts, _ = td.split_ranks_in_turn(path, L, rows, 30, 8, return_parts=True, timings=tm)
    assert np.array_equal(counts.astype(np.int64), want) and int(counts[:, 0].sum()) == 150 * n * m
    assert tm["one_sync_taken"] >= 8 and len(tm["rank_seconds"]) == 8 and min(tm["blocks_per_rank"]) > 4000
    assert tm["split_sub_taken"] == 8                               # (4 187 blocks a rank: two sub-ranges side by side, each rank)
    text1, counts1, _ = td.split_ranks_in_turn(path, L, rows, 30, 8, return_parts=True, split_sub=1)       # ... and every range in one piece: the same
    assert text1 == text and np.array_equal(counts1, counts)
    has, ins = orc.list_inserts(want, 30, lambda pos1: [])
    cons, _ = orc.build_consensus(30, want, [dict(o) for o in orfs], True, ins if has else None, True)
    assert text == orc.fasta_text("S", 30, cons)
    # (2)
    ref2, orfs2, reads2 = _consensus_case()
    want2, ins2 = _oracle_fasta(reads2, orfs2, len(ref2), 30)
    assert len(ins2) >= 4
    bamwriter.write_bam(path, reads2, "r", len(ref2), level=6, block=3000, split_records=True)
    rows2 = [{"start": o["start"], "end": o["end"], "strand": "+"} for o in orfs2]
    for sub_ranges in (1, 2, 3):                                     # the ranks' ranges in one piece / as sub-ranges side by side (their entries merged in file order)
        tm2 = {}
        text2, counts2, toks2 = td.split_ranks_in_turn(path, len(ref2), rows2, 30, 3, return_parts=True, split_sub=sub_ranges, timings=tm2)
        assert text2 == want2 and np.array_equal(counts2, c_oracle.tally(reads2, len(ref2))) and sum(1 for v in toks2.values() if v) >= 4, sub_ranges
        assert tm2["split_sub_taken"] == (3 if sub_ranges > 1 else 0), (sub_ranges, tm2)


def test_check_range_anchors_joins_the_ranks_ranges_into_one_chain():
    """distributed.check_range_anchors (what tally_split_bamfile's ranks agree on before their collective; tcmi_split_step carries the
    same condition as a telescoping sum): every range's first record must start where the range in front says its last one ends."""
    from trueconsense_amd.distributed import check_range_anchors as chk
    total = 10_000
    assert chk([(0, 4, -1, 2500), (4, 4, 2500, 5200), (8, 4, 5200, total)], total) is None
    assert chk([(0, 12, -1, -1)], total) is None                               # one range: the whole file, nothing to join
    assert chk([(0, 4, -1, 2500), (4, 0, -1, -1), (4, 8, 2500, total)], total) is None        # a rank without blocks
    assert chk([(0, 4, -1, 2500), (4, 4, -1, -1), (8, 4, 2500, total)], total) is None        # a range inside one long record: no start of its own
    assert chk([(0, 1, -1, 300), (1, 11, 300, total)], total) is None                         # rank 0: header blocks only, the header says where records begin
    why = chk([(0, 4, -1, 2500), (4, 4, 2466, 5200), (8, 4, 5200, total)], total)
    assert why and "rank 1" in why and "2466" in why and "2500" in why
    why = chk([(0, 4, -1, 2500), (4, 4, 2500, 5200), (8, 4, 5200, total - 7)], total)
    assert why and "last alignment record" in why
    why = chk([(0, 4, -1, -1), (4, 8, 2500, total)], total)                                   # nobody vouches for rank 1's start
    assert why and "no range in front" in why


def test_check_range_anchors_on_random_chains_and_cuts():
    """The joining rule on random record chains cut into random BGZF blocks and random block ranges: the true anchors always join,
    and a range whose first record start is off by any amount is always caught (as is a last record that does not end with the stream)."""
    from trueconsense_amd.distributed import check_range_anchors as chk
    rng = np.random.default_rng(123)
    for _ in range(300):
        header = int(rng.integers(20, 3000))
        sizes = rng.integers(40, 900, int(rng.integers(1, 400)))
        starts = header + np.concatenate(([0], np.cumsum(sizes)[:-1]))          # record starts in the stream
        total = int(header + sizes.sum())
        cuts = [0]
        while cuts[-1] < total:
            cuts.append(min(total, cuts[-1] + int(rng.integers(200, 2500))))   # block boundaries (records straddle them)
        nb = len(cuts) - 1
        world = int(rng.integers(1, 9))
        edges = sorted(set([0, nb] + [int(x) for x in rng.integers(0, nb + 1, world - 1)]))
        ranges = []
        for b0, b1 in zip(edges[:-1], edges[1:]):
            lo, hi = cuts[b0], cuts[b1]
            inside = starts[(starts >= lo) & (starts < hi)]
            first = int(inside[0]) if len(inside) and b0 != 0 else -1           # (the range that starts with the file is vouched for by the header)
            if len(inside):
                after = starts[starts >= hi]
                nxt = int(after[0]) if len(after) else total
            else:
                nxt = header if b0 == 0 and hi <= header else -1                # header blocks only: the header says where the records begin
            ranges.append((b0, b1 - b0, first, nxt))
        assert chk(ranges, total) is None, ranges
        cand = [i for i, r in enumerate(ranges) if r[2] >= 0]
        if cand:
            i = int(rng.choice(cand))
            b0, n, first, nxt = ranges[i]
            bad = list(ranges)
            bad[i] = (b0, n, first + int(rng.choice([-7, -1, 1, 33])), nxt)
            assert chk(bad, total) is not None, (ranges, i)
        last = max(i for i, r in enumerate(ranges) if r[3] >= 0)
        bad = list(ranges)
        bad[last] = ranges[last][:3] + (ranges[last][3] + 5,) if ranges[last][3] == total else bad[last]
        if bad[last] != ranges[last]:
            assert chk(bad, total) is not None


def test_split_reduce_hook_falls_back_to_the_process_groups_reduce(monkeypatch):
    """distributed.split_reduce_hook (what the command line's split worker and bench.py both call): where the RCCL hook library cannot be
    loaded — or a communicator cannot be made — the answer is torch.distributed's reduce, with the reason in words; `rccl=False`
    (a rehearsal of several ranks on one GPU) never tries.  No GPU: the hook library's loader is made to fail."""
    from trueconsense_amd import _ffi
    from trueconsense_amd import distributed as td

    def no_lib():
        raise ImportError("libtcmi_rccl.so is missing (test)")
    monkeypatch.setattr(_ffi, "rccl_lib", no_lib)
    comm, user, what = td.split_reduce_hook(0, 1)
    assert comm is None and user is None and "torch.distributed reduce" in what and "missing (test)" in what
    comm, user, what = td.split_reduce_hook(0, 1, rccl=False)
    assert comm is None and user is None and what == "torch.distributed reduce"
    monkeypatch.setenv("TCMI_SPLIT_HOOK", "torch")
    assert td.split_reduce_hook(0, 1)[2] == "torch.distributed reduce"
    td.split_reduce_hook_close(None)                                 # (nothing to destroy)
