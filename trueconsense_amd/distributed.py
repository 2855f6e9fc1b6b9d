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
import os

import numpy as np

from ._ffi import E_ARG as _ffi_E_ARG
from ._ffi import E_UNSUPPORTED as _ffi_E_UNSUPPORTED
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


def check_range_anchors(ranges, inflated_bytes):
    """ranges: per rank, in rank order, (first_block, n_blocks, first, next) — engine.ReadSet.range_anchors of its block range
    (tcmi_readset_range_anchors).  A range in the middle of the file starts at the first offset its first block finds PLAUSIBLE for an
    alignment record; that offset is a record start for sure only if the range in front ends there (by induction from the header,
    whose end the first range starts at).  -> None, or what does not join (a message): then no rank's result is to be trusted and
    the file takes the host reader."""
    expect = None                                                   # where the next record must start, once a range has fixed it
    for r, (first_block, n_blocks, first, nxt) in enumerate(ranges):
        if n_blocks <= 0:
            continue
        if first_block != 0 and first >= 0:
            if expect is not None and first != expect:
                return "rank %d's block range starts a record at stream offset %d, the range in front ends its last record at %d" % (r, first, expect)
            if expect is None and r > 0 and any(n > 0 for _, n, _, _ in ranges[:r]):
                return "rank %d's block range starts a record at stream offset %d, but no range in front says where its last record ends" % (r, first)
        if nxt >= 0:
            expect = nxt
    if expect is not None and expect != inflated_bytes:
        return "the last alignment record ends at stream offset %d, the file's stream at %d" % (expect, inflated_bytes)
    return None


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
        import torch.distributed as dist
        from ._ffi import TcmiError
        first, count = block_range(d.n_blocks, rank, world)
        # A range can be refused on ONE rank only (a record chain that does not close from a false start, a record longer than a block
        # at the range's end, a read the packer does not take): the ranks agree BEFORE anybody enters the collective — else the others
        # wait in it forever — and then all raise.
        rs, err = None, None
        try:
            rs = ctx.upload_bamfile(d, blocks=(first, count))
            if rs.max_end > L:
                err = (_ffi_E_ARG, "L=%d is smaller than the reads' extent %d" % (L, rs.max_end))
        except TcmiError as e:
            err = (e.code, str(e))
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            allv = [None] * dist.get_world_size(group)
            dist.all_gather_object(allv, (err, (first, count) + (rs.range_anchors if rs is not None else (-1, -1))), group=group)
            err = next((v[0] for v in allv if v[0]), None)
            if not err:                                             # ... and the ranges must join into ONE chain of records
                why = check_range_anchors([v[1] for v in allv], d.inflated_bytes)
                if why:
                    err = (_ffi_E_UNSUPPORTED, "%s: %s: host reader" % (path, why))
        if err:
            if rs is not None:
                rs.free()
            raise TcmiError(err[0], "tally_split_bamfile: a rank could not decode its block range: %s" % err[1])
        t = torch.zeros((7, ld), dtype=torch.int32, device="cuda")
        if rs.n_piled:
            check(lib().tcmi_tally_dev(ctx.handle, rs.handle, L, ld, C.c_void_p(t.data_ptr()), 0), ctx.handle)
        if to_root:
            reduce_counts(t, 0, group)
        else:
            allreduce_counts(t, group)
        mine = not to_root or not dist.is_initialized() or dist.get_rank(group) == 0
        counts = np.ascontiguousarray(t[:, :L].T.cpu().numpy()) if mine else None
        rs.free()
    finally:
        d.close()
        if own:
            ctx.close()
    return counts


# --------------------------------------------------------------------------------------------- configs[4], all the way to the consensus
def _entries_of_readset(ctx, rs, positions):
    """This rank's entries of the candidate columns (tcmi_readset_ins_entries) -> (bytes, ent_off list, long-insertion text bytes)."""
    from .engine import DEFAULT_FLAG_FILTER
    pos = np.ascontiguousarray(positions, np.int64)
    n = len(pos)
    off = np.zeros(n + 1, np.int64)
    cap, lcap = 1 << 14, 1 << 16
    while True:
        ents = np.zeros(cap * 48, np.uint8)
        text = np.zeros(lcap, np.uint8)
        used = C.c_int64(0)
        rc = lib().tcmi_readset_ins_entries(ctx.handle, rs.handle, n, pos.ctypes.data_as(C.c_void_p), DEFAULT_FLAG_FILTER, 1, ents.ctypes.data_as(C.c_void_p),
                                            cap, off.ctypes.data_as(C.c_void_p), text.ctypes.data_as(C.c_void_p), lcap, C.byref(used))
        if rc != 0 and (off[n] > cap or used.value > lcap):        # (the buffers were too small: the call says what is needed)
            cap, lcap = max(cap, int(off[n]) + 16), max(lcap, used.value + 16)
            continue
        check(rc, ctx.handle)
        return ents[:int(off[n]) * 48].tobytes(), off.tolist(), text[:used.value].tobytes()


def _vote(positions, pieces):
    """Rank 0: per column the ranks' pieces concatenated in rank order (= file order), voted on as tcmi_readset_modal_tokens votes
    (tcmi_modal_from_entries: pysam's default filters, Events.py:66).  -> ({pos: token or None}, status flags)."""
    from .engine import DEFAULT_MAX_DEPTH, DEFAULT_MIN_BASE_QUALITY
    n = len(positions)
    text = b"".join(p[2] for p in pieces)
    base = np.cumsum([0] + [len(p[2]) for p in pieces])
    cols, off = [], [0]
    arrays = []
    for r, (ents, eoff, _) in enumerate(pieces):
        a = np.frombuffer(ents, np.uint8).copy()
        if len(a) and base[r]:
            check(lib().tcmi_ins_entries_rebase(a.ctypes.data_as(C.c_void_p), len(a) // 48, int(base[r])))
        arrays.append(a)
    for k in range(n):
        for r, (_, eoff, _) in enumerate(pieces):
            cols.append(arrays[r][eoff[k] * 48:eoff[k + 1] * 48])
        off.append(off[-1] + sum(p[1][k + 1] - p[1][k] for p in pieces))
    allents = np.ascontiguousarray(np.concatenate(cols)) if cols and sum(len(c) for c in cols) else np.zeros(48, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        toff, cnt, st = np.zeros(n + 1, np.int64), np.zeros(n, np.int64), C.c_int32(0)
        tb = np.frombuffer(text, np.uint8) if text else np.zeros(1, np.uint8)
        rc = lib().tcmi_modal_from_entries(n, allents.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), DEFAULT_MIN_BASE_QUALITY, DEFAULT_MAX_DEPTH, 1,
                                           tb.ctypes.data_as(C.c_void_p), len(text), C.cast(buf, C.c_void_p), cap, toff.ctypes.data_as(C.c_void_p),
                                           cnt.ctypes.data_as(C.c_void_p), C.byref(st))
        if rc != 0 and b"token buffer too small" in (lib().tcmi_last_error(None) or b"") and cap < (1 << 30):
            cap *= 16
            continue
        check(rc)
        break
    return {int(positions[k]): (buf.raw[toff[k]:toff[k + 1]].decode("ascii") if cnt[k] else None) for k in range(n)}, st.value


class _Tokens:
    """What Events.inserts_from_flags asks a `bam` for (a token per candidate column)."""
    def __init__(self, toks):
        self.toks = toks

    def modal_token(self, pos):
        return self.toks.get(pos)


def rccl_communicator(rank, world, group=None):
    """An RCCL communicator of this process's own (include/tcmi_rccl.h) for tcmi_split_step's C hook: rank 0 draws the id, the bytes
    travel through the process group that is already up (any backend) — or nowhere at world 1.  -> (comm handle, _ffi.RcclUser);
    destroy with _ffi.rccl_lib().tcmi_rccl_comm_destroy(comm).  The current HIP device must be the rank's."""
    from . import _ffi
    # The ranks move in step: every rank enters the broadcast of the id and the gather of the verdicts whatever happened to it before
    # (a hook library that does not load, an id that cannot be drawn, a communicator that cannot be made) — and then all raise, or none.
    r, err = None, None
    try:
        r = _ffi.rccl_lib()
    except (ImportError, OSError) as e:
        err = str(e)
    ident = C.create_string_buffer(_ffi.RCCL_ID_BYTES)
    if rank == 0 and err is None and r.tcmi_rccl_unique_id(ident):
        err = (r.tcmi_rccl_last_error() or b"").decode()
    multi = world > 1
    if multi:
        import torch.distributed as dist
        box = [(err, ident.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        root_err, raw = box[0]
        if root_err is not None:
            raise RuntimeError("rccl_communicator: rank 0 could not draw the id: %s" % root_err)
        ident = C.create_string_buffer(raw, _ffi.RCCL_ID_BYTES)
    elif err is not None:
        raise RuntimeError("rccl_communicator: %s" % err)
    comm = C.c_void_p()
    if err is None and r.tcmi_rccl_comm_init(int(world), int(rank), ident, C.byref(comm)):
        err = (r.tcmi_rccl_last_error() or b"").decode()
        comm = C.c_void_p()
    if multi:
        allv = [None] * world
        dist.all_gather_object(allv, err, group=group)
        bad = [(k, v) for k, v in enumerate(allv) if v is not None]
        if bad:
            if comm:
                r.tcmi_rccl_comm_destroy(comm)
            raise RuntimeError("rccl_communicator: rank %d: %s" % bad[0])
    elif err is not None:
        raise RuntimeError("rccl_communicator: %s" % err)
    return comm, _ffi.RcclUser(comm, 0)


def split_reduce_hook(rank, world, group=None, rccl=True):
    """The exchange tcmi_split_step's hook runs on for this job -> (comm or None, rccl_user or None, what it is, in words).
    rccl: the C hook tcmi_rccl_reduce (libtcmi_rccl.so: ncclReduce queued on the context's stream) on a communicator of this job's own —
    at ONE rank too, so that the collective itself runs wherever the step runs.  All ranks get the same answer (rccl_communicator's
    ranks raise together): where the communicator cannot be made — no libtcmi_rccl.so, several ranks on one GPU — every rank falls
    back to torch.distributed's reduce on the process group that is already up.  Destroy `comm` with split_reduce_hook_close."""
    if rccl and os.environ.get("TCMI_SPLIT_HOOK", "rccl") == "rccl":
        try:
            comm, user = rccl_communicator(rank, world, group)
            return comm, user, "tcmi_rccl_reduce: ncclReduce on the context's stream, %d-rank communicator" % world
        except RuntimeError as e:
            return None, None, "torch.distributed reduce (an RCCL communicator of its own could not be made: %s)" % e
    return None, None, "torch.distributed reduce"


def split_reduce_hook_close(comm):
    if comm:
        from . import _ffi
        _ffi.rccl_lib().tcmi_rccl_comm_destroy(comm)


def consensus_split_bamfile(path, ref_len, gff_rows, mincov, include_ambig=True, name="S", rank=0, world=1, device=0, group=None,
                            step_fn=None, entries_fn=None, return_parts=False, rccl_user=None, ctx=None, dbam=None, timings=None):
    """BASELINE configs[4] all the way: ONE BAM file over `world` ranks -> its consensus FASTA text on rank 0 (None on the others).
    What the ranks jointly replace is the reference's single pile-up pass (indexing.py:96-100), its insert candidates' region
    pile-ups (Events.py:47-82) and the walk (Sequences.py:168-322):

      1. every rank decodes, packs and tallies the records that start in ITS contiguous range of the file's BGZF blocks and the count
         matrices are summed to rank 0 (tcmi_split_step: the step in C, the reduce a hook — with `rccl_user` (rccl_communicator) the C
         hook tcmi_rccl_reduce: ncclReduce queued on the context's stream; else torch.distributed's reduce, RCCL under "nccl"); a rank
         that cannot decode its range still takes part (zeros + a failure word): nobody waits for it forever, and every rank learns
         the verdict;
      2. rank 0 calls (HIP call kernel) and broadcasts the insert-candidate columns;
      3. every rank collects the 48-byte entries of those columns from its resident stream (ins_entries_kernel); the ranks agree that
         all of them could (a rank whose range holds reads the entry kernel does not take — longer than 512 positions — says so
         instead of leaving the others in the gather), then send them to rank 0;
      4. rank 0 concatenates the pieces per column in rank order — file order: pysam's max_depth admission and first-seen tie-break
         depend on it —, votes (tcmi_modal_from_entries), and walks.  Overlapping mates whose other mate would have to be looked at on
         another rank, or a rank that could not collect its entries: rank 0 decodes the file on the host for the tokens
         (tcmi_bam_load + tcmi_modal_tokens).

    step_fn(first, count) -> (counts int [L,7] of the rank's range) and entries_fn(positions) -> (entry bytes, ent_off, long text)
    replace the GPU on boxes without one (tests: the oracle's tally and entries; the call then comes from the oracle too).
    return_parts: rank 0 gets (FASTA text, counts int32 [L,7], {candidate column: modal token}) — what the command line's other
    writers need (Outputs.WriteOutputs, Coverage.BuildCoverage).  ctx / dbam: the caller's (kept open: a bench loop); timings: a dict
    that receives the seconds of "step" (tcmi_split_step, reduce included) and "entries" (steps 2-4 up to the vote)."""
    import time
    import torch
    import torch.distributed as dist
    from ._ffi import TcmiError
    from .Events import inserts_from_flags
    from .Sequences import consensus_from_records
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    L = int(ref_len)
    ld = (L + 255) // 256 * 256
    root = rank == 0
    plain = alt = flags = counts_root = None
    own_ctx, own_d = ctx is None, dbam is None
    d, rs = dbam, None
    err = None
    t0 = time.perf_counter()
    try:
        if step_fn is None:
            from .engine import Context, DeviceBam, ReadSet
            torch.cuda.set_device(device)
            try:
                if ctx is None:
                    ctx = Context(device, stream=torch.cuda.current_stream().cuda_stream)
                if d is None:
                    d = DeviceBam(path)
            except (TcmiError, OSError) as e:                        # this rank cannot even start: it must still meet the others in the reduce
                err = (getattr(e, "code", _ffi_E_ARG), str(e))
            n_tail = 6 * int(world) + 1                              # TCMI_SPLIT_TAIL_WORDS(world)
            t = torch.zeros(7 * ld + n_tail, dtype=torch.int32, device="cuda")
            if err is not None:
                t[-1] = 1
                if rccl_user is not None:                            # (the others are in ncclReduce on that communicator)
                    from . import _ffi
                    rrc = _ffi.rccl_lib().tcmi_rccl_reduce(C.cast(C.pointer(rccl_user), C.c_void_p), C.c_void_p(t.data_ptr()), t.numel(),
                                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
                    torch.cuda.synchronize()
                    if rrc:                                          # (the failure this rank reports gains the reduce's own)
                        err = (err[0], "%s; and its share of the reduce failed too: %s" % (err[1], (_ffi.rccl_lib().tcmi_rccl_last_error() or b"").decode()))
                else:
                    reduce_counts(t, 0, group)
            else:
                first, count = block_range(d.n_blocks, rank, world)

                @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
                def hook(user, ptr, n, stream):                      # (the tensor IS the buffer at `ptr`; torch's collective runs on the context's stream)
                    try:
                        reduce_counts(t, 0, group)
                        return 0
                    except Exception:                                # noqa: BLE001 — a C caller gets a code, not an exception
                        return 1
                if rccl_user is not None:
                    from . import _ffi
                    fn, user = C.cast(_ffi.rccl_lib().tcmi_rccl_reduce, C.c_void_p), C.cast(C.pointer(rccl_user), C.c_void_p)
                else:
                    fn, user = hook, None
                h, p_, a_, f_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
                rc = lib().tcmi_split_step(ctx.handle, d.handle, first, count, L, ld, C.c_void_p(t.data_ptr()), int(mincov), int(bool(include_ambig)), fn, user,
                                           int(rank), int(world), C.byref(h), C.byref(p_), C.byref(a_), C.byref(f_))
                if rc:
                    err = (rc, (lib().tcmi_last_error(ctx.handle) or b"").decode("utf-8", "replace"))
                else:
                    rs = ReadSet(ctx, h, None)
                    if root:
                        def grab(vp):
                            out = np.empty(L, np.uint8)
                            C.memmove(out.ctypes.data, vp, L)
                            return out
                        plain, alt, flags = grab(p_), grab(a_), grab(f_)
                        if return_parts:
                            counts_root = np.ascontiguousarray(t[:7 * ld].view(7, ld)[:, :L].T.cpu().numpy())
        else:
            d_blocks = step_fn("n_blocks")
            first, count = block_range(d_blocks, rank, world)
            part = torch.from_numpy(np.ascontiguousarray(np.asarray(step_fn((first, count))).T.astype(np.int32)))
            if multi:
                dist.reduce(part, dst=0, op=dist.ReduceOp.SUM, group=group)
            if root:
                counts_root = np.ascontiguousarray(part.numpy().T)
                plain, alt, flags = step_fn(("call", counts_root, int(mincov), bool(include_ambig)))
        if timings is not None:
            timings["step"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        # every rank learns whether the step held everywhere (a failed rank took part in the reduce: no deadlock, but no consensus either)
        verdict = [err]
        if multi:
            allv = [None] * dist.get_world_size(group)
            dist.all_gather_object(allv, err, group=group)
            verdict = allv
        bad = [v for v in verdict if v]
        if bad:
            raise TcmiError(bad[0][0], "consensus_split_bamfile: %s" % bad[0][1])
        # the insert-candidate columns, from rank 0 to everybody
        cand = [(np.nonzero(flags & 8)[0] + 1).tolist()] if root else [None]
        if multi:
            dist.broadcast_object_list(cand, src=0, group=group)
        cand = cand[0]
        text = None
        host_sweep = False
        if cand:
            piece, perr = None, None
            try:
                piece = entries_fn(cand) if entries_fn is not None else _entries_of_readset(ctx, rs, cand)
            except TcmiError as e:                                   # e.g. TCMI_E_UNSUPPORTED: a read of more than 512 positions in this rank's range
                perr = (e.code, str(e))
            pieces = [piece]
            if multi:                                                # nobody enters the gather unless everybody has a piece
                allp = [None] * dist.get_world_size(group)
                dist.all_gather_object(allp, perr, group=group)
                perr = next((v for v in allp if v), None)
            if perr is not None:
                if perr[0] != _ffi_E_UNSUPPORTED:
                    raise TcmiError(perr[0], "consensus_split_bamfile: insert entries: %s" % perr[1])
                host_sweep = True                                    # rank 0 sweeps the file on the host, as it does for st & 2 below
            elif multi:
                pieces = [None] * dist.get_world_size(group) if root else None
                dist.gather_object(piece, pieces, dst=0, group=group)
        toks = {}
        if root:
            if cand:
                st = 2
                if not host_sweep:
                    toks, st = _vote(cand, pieces)
                if st & 2:                                          # a pair of mates that only a sweep over the whole file resolves
                    from .engine import BamFile, modal_tokens
                    bam = BamFile(path)
                    try:
                        toks = {p: tk for p, (tk, _) in modal_tokens(bam, cand).items()}
                    finally:
                        bam.close()
            if timings is not None:
                timings["entries"] = time.perf_counter() - t1
            _, inserts = inserts_from_flags(flags, _Tokens(toks))
            gff = {i: dict(r) for i, r in enumerate(gff_rows)}
            cons, _ = consensus_from_records(plain, alt, flags, gff, inserts, True)
            text = ">%s mincov=%d\n%s\n" % (name, int(mincov), cons)
    finally:
        t2 = time.perf_counter()
        if rs is not None:
            rs.free()
        if d is not None and own_d:
            d.close()
        if ctx is not None and own_ctx and step_fn is None:
            ctx.close()
        if timings is not None:
            timings["release"] = time.perf_counter() - t2
            timings["total"] = time.perf_counter() - t0
    if return_parts:
        return (text, counts_root, toks) if root else None
    return text


def split_ranks_in_turn(path, ref_len, gff_rows, mincov, world, include_ambig=True, name="S", device=0, return_parts=False, timings=None,
                        split_sub=None):
    """BASELINE configs[4] at ANY world size on the ONE GPU there is: the ranks' steps of a `world`-GPU job played one after the other
    on one context — rank world-1 first, rank 0 (the root) last — each through tcmi_split_step exactly as a rank of the real job runs
    it (its own contiguous range of the file's BGZF blocks + the block behind it, the range table and the failure word behind the
    matrix, the root's pairwise check of the joins, the call kernel on the root).  The hook stands in for the collective: a non-root
    rank's buffer is added to an accumulator on the device, the root's hook adds the accumulator to its own — the integer sum RCCL's
    reduce delivers (indexing.py:96-100 is the pass the ranks share).  What it cannot show is the collective's time: `timings`
    receives per rank the seconds of its step ("rank_seconds"), so that a node's step can be projected as max(rank) + reduce.
    Insert candidates are voted on from every rank's entries, gathered in rank order as consensus_split_bamfile's step 4 does
    (Events.py:47-82).  -> FASTA text (return_parts: + counts int32 [L,7] + {column: token})."""
    import time
    import torch
    from ._ffi import TcmiError
    from .engine import Context, DeviceBam, ReadSet
    from .Events import inserts_from_flags
    from .Sequences import consensus_from_records
    L = int(ref_len)
    ld = (L + 255) // 256 * 256
    world = int(world)
    torch.cuda.set_device(device)
    ctx = Context(device, stream=torch.cuda.current_stream().cuda_stream)
    if split_sub is not None:                                        # (tcmi_split_step's sub-ranges per rank: 0 = auto, 1 = never)
        ctx.set_option("split_sub", int(split_sub))
    d = DeviceBam(path)
    n_words = 7 * ld + 6 * world + 1                                 # TCMI_SPLIT_TAIL_WORDS(world)
    acc = torch.zeros(n_words, dtype=torch.int32, device="cuda")
    t = torch.zeros(n_words, dtype=torch.int32, device="cuda")
    state = {"root": False}

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    def hook(user, ptr_, n, stream):                                 # (ptr_ IS t's buffer; the context runs on torch's current stream)
        if int(n) != n_words or int(ptr_) != t.data_ptr():
            return 1
        if state["root"]:
            t.add_(acc)
        else:
            acc.add_(t)
        return 0

    secs, sets = [0.0] * world, [None] * world
    plain = alt = flags = None
    try:
        for rank in range(world - 1, -1, -1):
            first, count = block_range(d.n_blocks, rank, world)
            state["root"] = rank == 0
            h, p_, a_, f_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = lib().tcmi_split_step(ctx.handle, d.handle, first, count, L, ld, C.c_void_p(t.data_ptr()), int(mincov), int(bool(include_ambig)), hook, None,
                                       rank, world, C.byref(h), C.byref(p_), C.byref(a_), C.byref(f_))
            torch.cuda.synchronize()
            secs[rank] = time.perf_counter() - t0
            if rc:
                raise TcmiError(rc, "split_ranks_in_turn: rank %d of %d: %s" % (rank, world, (lib().tcmi_last_error(ctx.handle) or b"").decode("utf-8", "replace")))
            sets[rank] = ReadSet(ctx, h, None)
            if rank == 0:
                grab = lambda vp: np.frombuffer(C.string_at(vp, L), np.uint8).copy()
                plain, alt, flags = grab(p_), grab(a_), grab(f_)
            elif rank > 0:                                           # (a context holds ONE resident stream: this rank's entries are collected below, from a decode of their own)
                sets[rank].free()
                sets[rank] = None
        counts = np.ascontiguousarray(t[:7 * ld].view(7, ld)[:, :L].T.cpu().numpy()) if return_parts else None
        cand = (np.nonzero(flags & 8)[0] + 1).tolist()
        toks = {}
        if cand:
            pieces = []
            for rank in range(world):                                # rank order = file order
                rs = sets[rank]
                if rs is None:
                    rs = ctx.upload_bamfile(d, blocks=block_range(d.n_blocks, rank, world))
                try:
                    pieces.append(_entries_of_readset(ctx, rs, cand))
                finally:
                    rs.free()
                    sets[rank] = None
            toks, st = _vote(cand, pieces)
            if st & 2:
                from .engine import BamFile, modal_tokens
                bam = BamFile(path)
                try:
                    toks = {p: tk for p, (tk, _) in modal_tokens(bam, cand).items()}
                finally:
                    bam.close()
        _, inserts = inserts_from_flags(flags, _Tokens(toks))
        cons, _ = consensus_from_records(plain, alt, flags, {i: dict(r) for i, r in enumerate(gff_rows)}, inserts, True)
        text = ">%s mincov=%d\n%s\n" % (name, int(mincov), cons)
        if timings is not None:
            timings.update(rank_seconds=secs, blocks_per_rank=[block_range(d.n_blocks, r, world)[1] for r in range(world)],
                           n_blocks=d.n_blocks, file_bytes=d.file_bytes, inflated_bytes=d.inflated_bytes,
                           decode_batched=ctx.stat("decode_batched"), one_sync_taken=ctx.stat("one_sync_taken"), split_sub_taken=ctx.stat("split_sub_taken"))
    finally:
        for rs in sets:
            if rs is not None:
                rs.free()
        d.close()
        ctx.close()
    return (text, counts, toks) if return_parts else text
