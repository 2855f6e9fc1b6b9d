"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Two shard shapes (BASELINE.json configs[3] and [4]):

* many BAMs  — independent objects: rank r takes BAMs r, r+world, ...; NO data-path collective.
* one BAM    — the tally is a commutative integer sum: every rank takes a contiguous range of the FILE's
               BGZF blocks (tally_split_bamfile: it maps the same file, sends only its range's compressed
               bytes to its GPU, decodes and packs the records that start in those blocks there) — or, for
               reads already decoded, a contiguous range of the read arrays (tally_split_bam) —, tallies
               it into a full-length int32 [7][ld] matrix in HBM, and ONE exchange of that matrix
               (837 284 B at L = 29 903) follows: a reduce to the rank that calls and walks
               (reduce_counts), or an all-reduce when every rank wants the whole matrix (allreduce_counts).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._ffi import check, lib


def shard_items(n_items, rank, world):
    """Indices of the independent BAMs rank `rank` processes (round-robin)."""
    return list(range(rank, n_items, world))


def read_range(n_reads, rank, world):
    """Contiguous read range [a, b) of rank `rank`; ranges partition [0, n_reads)."""
    base, rem = divmod(int(n_reads), int(world))
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def block_range(n_blocks, rank, world):
    """BGZF blocks [first, first + count) of rank `rank`: contiguous, the ranges partition the file.  (Header and end-of-file
    blocks are simply part of some range: they hold no records.)"""
    a, b = read_range(n_blocks, rank, world)
    return a, b - a


def shard_reads(reads, rank, world):
    """The rank's contiguous slice of a dict of flat read arrays (tcmi_reads layout); a read that
    straddles a boundary simply belongs to one range."""
    n = int(reads["n_reads"])
    a, b = read_range(n, rank, world)
    co, so = np.asarray(reads["cigar_off"]), np.asarray(reads["seq_off"])
    out = {"n_reads": b - a, "pos": reads["pos"][a:b], "flag": reads["flag"][a:b], "l_qseq": reads["l_qseq"][a:b],
           "cigar_off": (co[a:b + 1] - co[a]).astype(np.uint64), "cigar": reads["cigar"][int(co[a]):int(co[b])],
           "seq_off": (so[a:b + 1] - so[a]).astype(np.uint64), "seq": reads["seq"][int(so[a]):int(so[b])]}
    if reads.get("tid") is not None:
        out["tid"] = reads["tid"][a:b]
    if reads.get("qual") is not None:
        lq = np.asarray(reads["l_qseq"], np.int64)
        q0 = int(lq[:a].sum())
        out["qual"] = reads["qual"][q0:q0 + int(lq[a:b].sum())]
    return out


def allreduce_counts(t, group=None):
    """In-place sum of the [7][ld] int32 matrix over the ranks.  A CUDA tensor goes through the
    process group's own backend (RCCL when it is "nccl"); under gloo it is staged through the host."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_counts(t, dst=0, group=None):
    """Sum of the [7][ld] int32 matrix over the ranks, delivered to rank `dst` only (the call kernel and the walk
    run on one GPU, so nobody else needs the whole matrix: SURVEY §8-e).  RCCL reduce under "nccl"; staged
    through the host under gloo.  The other ranks' tensors are left as they were."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) == dst:
            t.copy_(h)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return t


def tally_split_bam(reads, L, rank, world, device=0, group=None, tally_fn=None):
    """BASELINE configs[4]: one BAM over `world` ranks -> the whole int32 [L,7] matrix on every rank.

    tally_fn(shard, L) -> int [L,7] replaces the GPU tally (tests on CPU-only boxes pass the
    oracle's; the product path leaves it None and needs an MI355X)."""
    import torch
    shard = shard_reads(reads, rank, world)
    L = int(L)
    if tally_fn is not None:
        part = torch.from_numpy(np.ascontiguousarray(np.asarray(tally_fn(shard, L)).T.astype(np.int32)))   # [7][L]
        allreduce_counts(part, group)
        return np.ascontiguousarray(part.numpy().T)
    from .engine import Context
    ld = (L + 255) // 256 * 256
    torch.cuda.set_device(device)
    t = torch.zeros((7, ld), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream                 # tally and collective on torch's stream
    ctx = Context(device, stream=stream)
    rs = ctx.upload(shard)
    check(lib().tcmi_tally_dev(ctx.handle, rs.handle, L, ld, C.c_void_p(t.data_ptr()), 0), ctx.handle)
    allreduce_counts(t, group)
    counts = np.ascontiguousarray(t[:, :L].T.cpu().numpy())
    rs.free()
    ctx.close()
    return counts


def tally_split_bamfile(path, L, rank, world, device=0, group=None, ctx=None, to_root=False):
    """BASELINE configs[4] from ONE FILE: every rank opens the same BAM, decodes on its GPU only the alignment records that start
    in its contiguous range of BGZF blocks (engine.Context.upload_bamfile(blocks=...): inflate, record index, pack), tallies
    them into a full-length int32 [7][ld] matrix and the matrices are summed — to every rank, or (to_root) to rank 0 only.
    -> int [L,7] (None on the other ranks with to_root).  No rank ever holds the file's reads, decoded or not."""
    import torch
    from .engine import Context, DeviceBam
    L = int(L)
    ld = (L + 255) // 256 * 256
    torch.cuda.set_device(device)
    own = ctx is None
    if own:
        ctx = Context(device, stream=torch.cuda.current_stream().cuda_stream)         # tally and collective on torch's stream
    d = DeviceBam(path)
    try:
        first, count = block_range(d.n_blocks, rank, world)
        rs = ctx.upload_bamfile(d, blocks=(first, count))
        t = torch.zeros((7, ld), dtype=torch.int32, device="cuda")
        if rs.n_piled:
            check(lib().tcmi_tally_dev(ctx.handle, rs.handle, L, ld, C.c_void_p(t.data_ptr()), 0), ctx.handle)
        if to_root:
            reduce_counts(t, 0, group)
        else:
            allreduce_counts(t, group)
        import torch.distributed as dist
        mine = not to_root or not dist.is_initialized() or dist.get_rank(group) == 0
        counts = np.ascontiguousarray(t[:, :L].T.cpu().numpy()) if mine else None
        rs.free()
    finally:
        d.close()
        if own:
            ctx.close()
    return counts
