"""Device context and the host-side calls of libtcmi, as thin Python objects.

    Context      one HIP stream on one MI355X (tcmi_ctx)
    ReadSet      reads resident in HBM (tcmi_readset)
    BamFile      a decoded BAM (tcmi_bam): flat read arrays + header
    consensus_walk / modal_tokens   the HOST entry points (no GPU involved)

Everything that computes goes through the C ABI (include/tcmi.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, lib, ptr


class ReadSet:
    def __init__(self, ctx, handle, keep):
        self.ctx, self.handle, self._keep = ctx, handle, keep
        v = [C.c_int64(0) for _ in range(5)]
        check(lib().tcmi_readset_info(handle, *[C.byref(x) for x in v]))
        self.n_reads, self.n_piled, self.algorithmic_bytes, self.device_bytes, self.max_end = (x.value for x in v)
        o = C.c_int32(0)
        check(lib().tcmi_readset_origin(handle, C.byref(o)))
        self.packed_on_device = bool(o.value)       # pack_device.hip built it (else the host packer)
        a, b = C.c_int64(-1), C.c_int64(-1)
        check(lib().tcmi_readset_range_anchors(handle, C.byref(a), C.byref(b)))
        # a block range of a file: where its first record starts (a range in the middle of the file; nothing in front vouches for it)
        # and where the first record behind it starts, offsets into the file's inflated stream (-1: none) — distributed.check_range_anchors
        self.range_anchors = (a.value, b.value)

    def free(self):
        if self.handle:
            lib().tcmi_readset_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device + stream.  Raises TcmiError(E_NODEVICE) when no gfx950 GPU is usable."""

    def __init__(self, device=0, stream=None):
        """stream: a hipStream_t as an int (0 = the default stream) to run on, e.g. another Context's
        `.stream` or torch.cuda.current_stream().cuda_stream; None = a stream of its own."""
        h = C.c_void_p()
        if stream is None:
            check(lib().tcmi_ctx_create(int(device), C.byref(h)))
        else:
            check(lib().tcmi_ctx_create_on_stream(int(device), C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            lib().tcmi_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing
    def sync(self):
        check(lib().tcmi_ctx_sync(self.handle), self.handle)

    @property
    def stream(self):
        return lib().tcmi_ctx_stream(self.handle)

    def set_option(self, key, value):
        check(lib().tcmi_ctx_set_option(self.handle, key.encode(), int(value)), self.handle)

    def stat(self, key):
        v = C.c_int64(0)
        check(lib().tcmi_ctx_stat(self.handle, key.encode(), C.byref(v)), self.handle)
        return v.value

    def profile(self, on=True):
        check(lib().tcmi_profile_enable(self.handle, int(on)), self.handle)
        check(lib().tcmi_profile_reset(self.handle), self.handle)

    def profile_get(self, kernel):
        ms, n = C.c_double(0), C.c_int64(0)
        check(lib().tcmi_profile_get(self.handle, kernel, C.byref(ms), C.byref(n)), self.handle)
        return ms.value, n.value

    # ---- stage A
    def upload(self, reads):
        """reads: dict of flat arrays (tcmi_reads layout) or BamFile -> ReadSet in HBM."""
        r, keep = reads.as_struct() if isinstance(reads, BamFile) else _ffi.as_reads(reads)
        h = C.c_void_p()
        check(lib().tcmi_readset_upload(self.handle, C.byref(r), C.byref(h)), self.handle)
        return ReadSet(self, h, keep)

    def upload_batch(self, reads_list, stride):
        """Several BAMs in one read set: BAM b's positions are shifted by b * stride (multiple of 256)."""
        structs, keep = [], []
        for r in reads_list:
            st, k = r.as_struct() if isinstance(r, BamFile) else _ffi.as_reads(r)
            structs.append(st)
            keep.append(k)
        arr = (C.POINTER(_ffi.Reads) * len(structs))(*[C.pointer(s) for s in structs])
        h = C.c_void_p()
        check(lib().tcmi_readset_upload_batch(self.handle, arr, len(structs), int(stride), C.byref(h)), self.handle)
        return ReadSet(self, h, (keep, structs))

    def upload_bamfile(self, dbam, blocks=None):
        """DeviceBam -> ReadSet: BGZF inflate, record index and packing all on the device.  blocks = (first, count): only the
        records that start in that range of the file's BGZF blocks (ranks that share one file take a range each)."""
        h, n = C.c_void_p(), C.c_int64(0)
        first, count = (0, -1) if blocks is None else blocks
        check(lib().tcmi_readset_from_bamfile_blocks(self.handle, dbam.handle, int(first), int(count), C.byref(h), C.byref(n)), self.handle)
        rs = ReadSet(self, h, None)
        return rs

    def bamfile_step(self, dbam, ref_len, mincov, include_ambig, want_counts=True):
        """DeviceBam -> (ReadSet, plain, alt, flags, counts or None) with ONE wait of the host (tcmi_bamfile_step): decode, pack,
        tally and call queued back to back; a file the one-pass packer does not take goes through upload_bamfile + step."""
        h, L = C.c_void_p(), C.c_int64(0)
        p, a, f, c = (C.c_void_p() for _ in range(4))
        ld = C.c_int64(0)
        check(lib().tcmi_bamfile_step(self.handle, dbam.handle, int(ref_len), int(mincov), int(bool(include_ambig)), C.byref(h), C.byref(L),
                                      C.byref(p), C.byref(a), C.byref(f), C.byref(c) if want_counts else None, C.byref(ld)), self.handle)
        n = L.value

        def grab(vp, dt, k):
            out = np.empty(k, dt)
            C.memmove(out.ctypes.data, vp, out.nbytes)
            return out
        plain, alt, flags = grab(p, np.uint8, n), grab(a, np.uint8, n), grab(f, np.uint8, n)
        counts = None
        if want_counts:
            counts = np.ascontiguousarray(grab(c, np.int32, 7 * ld.value).reshape(7, ld.value)[:, :n].T)
        return ReadSet(self, h, None), plain, alt, flags, counts

    def readset_modal_tokens(self, readset, positions, min_base_quality=13, flag_filter=0x4 | 0x100 | 0x200 | 0x400,
                             ignore_orphans=True, max_depth=8000, ignore_overlaps=True):
        """modal_tokens() for a read set the device decoded: the reads of each candidate column are examined by a HIP kernel in
        the still-resident inflated stream.  Raises TcmiError(E_UNSUPPORTED) when that stream is gone or a token does not fit."""
        positions = np.ascontiguousarray(sorted(int(p) for p in positions), np.int64)
        n = len(positions)
        if n == 0:
            return {}
        cap = 1 << 16
        buf = C.create_string_buffer(cap)
        off, cnt, st = np.zeros(n + 1, np.int64), np.zeros(n, np.int64), C.c_int32(0)
        check(lib().tcmi_readset_modal_tokens(self.handle, readset.handle, n, ptr(positions), int(min_base_quality), int(flag_filter),
                                              int(bool(ignore_orphans)), int(max_depth), int(bool(ignore_overlaps)), C.cast(buf, C.c_void_p), cap,
                                              ptr(off), ptr(cnt), C.byref(st)), self.handle)
        if st.value & 2:
            raise _ffi.TcmiError(_ffi.E_UNSUPPORTED, "overlapping mates with a deletion on an insert-candidate column")
        return {int(positions[k]): (buf.raw[off[k]:off[k + 1]].decode("ascii") if cnt[k] else None, int(cnt[k])) for k in range(n)}

    def tally(self, reads, L=None, ref_len=0):
        """reads -> int32 [L,7] (coverage,A,T,C,G,X,I); L defaults to max(ref_len, read extent)."""
        r, keep = reads.as_struct() if isinstance(reads, BamFile) else _ffi.as_reads(reads)
        if L is None:
            L = reads_extent(reads, ref_len)
        counts = np.zeros((int(L), 7), np.int32)
        check(lib().tcmi_tally(self.handle, C.byref(r), int(L), ptr(counts)), self.handle)
        del keep
        return counts

    def tally_dev(self, readset, L, ld, d_counts, zero=True):
        """Device-resident tally into int32 [7][ld] planes at device address `d_counts` (e.g. a torch
        tensor's data_ptr()), on this context's stream."""
        check(lib().tcmi_tally_dev(self.handle, readset.handle, int(L), int(ld), C.c_void_p(int(d_counts)), int(bool(zero))),
              self.handle)

    def call_dev(self, d_counts, L, ld, mincov, include_ambig, d_plain, d_alt, d_flags):
        """Device-resident call: counts planes -> three uint8[ld] record planes (device addresses)."""
        check(lib().tcmi_call_dev(self.handle, C.c_void_p(int(d_counts)), int(L), int(ld), int(mincov), int(bool(include_ambig)),
                                  C.c_void_p(int(d_plain)), C.c_void_p(int(d_alt)), C.c_void_p(int(d_flags)), None, None),
              self.handle)

    # ---- stage B, position-local
    def call(self, counts, mincov, include_ambig, want_events=False):
        """counts [L,7] -> (plain, alt, flags) uint8 [L] (+ ascending 0-based event indices)."""
        counts = np.ascontiguousarray(counts, np.int32)
        L = len(counts)
        plain, alt, flags = (np.empty(L, np.uint8) for _ in range(3))
        ev = np.empty(L, np.int32) if want_events else None
        n_ev = C.c_int64(0)
        check(lib().tcmi_call(self.handle, ptr(counts), L, int(mincov), int(bool(include_ambig)), ptr(plain),
                              ptr(alt), ptr(flags), ptr(ev), C.byref(n_ev)), self.handle)
        if want_events:
            return plain, alt, flags, ev[:n_ev.value].copy()
        return plain, alt, flags

    # ---- whole resident step: zero + tally + call + records to pinned host memory
    def step(self, readset, L, mincov, include_ambig, want_counts=True):
        self.step_begin(readset, L, mincov, include_ambig, want_counts)
        return self.step_end()

    def step_begin(self, readset, L, mincov, include_ambig, want_counts=True):
        """Enqueue the step on this context's stream and return at once."""
        check(lib().tcmi_step_begin(self.handle, readset.handle, int(L), int(mincov), int(bool(include_ambig)),
                                    int(bool(want_counts))), self.handle)
        self._step = (int(L), bool(want_counts))

    def step_end(self):
        """Wait for the stream; -> (plain, alt, flags, counts or None) as fresh numpy arrays."""
        L, want_counts = self._step
        p, a, f, c = (C.c_void_p() for _ in range(4))
        ld = C.c_int64(0)
        check(lib().tcmi_step_end(self.handle, C.byref(p), C.byref(a), C.byref(f), C.byref(c), C.byref(ld)),
              self.handle)

        def grab(vp, dt, n):
            out = np.empty(n, dt)
            C.memmove(out.ctypes.data, vp, out.nbytes)
            return out
        plain, alt, flags = grab(p, np.uint8, L), grab(a, np.uint8, L), grab(f, np.uint8, L)
        counts = None
        if want_counts:
            planes = grab(c, np.int32, 7 * ld.value).reshape(7, ld.value)
            counts = np.ascontiguousarray(planes[:, :L].T)
        return plain, alt, flags, counts


# --------------------------------------------------------------------------- host-only entry points
def reads_extent(reads, ref_len=0):
    r, keep = reads.as_struct() if isinstance(reads, BamFile) else _ffi.as_reads(reads)
    out = C.c_int64(0)
    check(lib().tcmi_reads_extent(C.byref(r), int(ref_len), C.byref(out)))
    del keep
    return out.value


class WalkKeyError(KeyError):
    """The reference raises KeyError here (Sequences.py:47): a deletion walk ran past the last position."""


def consensus_walk(plain, alt, flags, orf_start, orf_end, orf_is_plus, ins_pos, ins_shift, ins_seqs,
                   include_ins):
    """Sequential part of BuildConsensus (Sequences.py:179-322 + ORFs.py) over call records.
    -> (consensus str, new_start int64[n_orf], new_end int64[n_orf])."""
    plain, alt, flags = (np.ascontiguousarray(x, np.uint8) for x in (plain, alt, flags))
    L = len(plain)
    os_, oe = np.ascontiguousarray(orf_start, np.int64), np.ascontiguousarray(orf_end, np.int64)
    op = np.ascontiguousarray(orf_is_plus, np.uint8)
    ip = np.ascontiguousarray(ins_pos, np.int64)
    ish = np.ascontiguousarray(ins_shift, np.int32)
    blob = "".join(ins_seqs).encode("ascii")
    off = np.zeros(len(ins_seqs) + 1, np.int64)
    if len(ins_seqs):
        off[1:] = np.cumsum([len(s) for s in ins_seqs])
    cap = L + len(blob) + 1
    out = C.create_string_buffer(cap)
    n_out, err = C.c_int64(0), C.c_int64(0)
    ns, ne = np.zeros(len(os_), np.int64), np.zeros(len(os_), np.int64)
    rc = lib().tcmi_consensus_walk(ptr(plain), ptr(alt), ptr(flags), L, len(os_), ptr(os_), ptr(oe), ptr(op),
                                   len(ip), ptr(ip), ptr(ish), blob, ptr(off), int(bool(include_ins)),
                                   C.cast(out, C.c_void_p), cap, C.byref(n_out), ptr(ns), ptr(ne), C.byref(err))
    if rc == _ffi.E_KEYERROR:
        raise WalkKeyError(err.value)
    if rc == _ffi.E_ZERODIV:
        raise ZeroDivisionError("division by zero")
    check(rc)
    return out.raw[:n_out.value].decode("ascii"), ns, ne


class Pipeline:
    """Native batch runner (tcmi_pipeline): many resident BAMs -> consensus sequences, GPU steps
    queued ahead on one stream, walks on `walkers` host threads."""

    def __init__(self, device=0, slots=4, walkers=4):
        h = C.c_void_p()
        check(lib().tcmi_pipeline_create(int(device), int(slots), int(walkers), C.byref(h)))
        self.handle, self.slots = h, int(slots)
        self.ctx = Context.__new__(Context)            # slot 0's context, owned by the pipeline
        self.ctx.handle, self.ctx.device = C.c_void_p(lib().tcmi_pipeline_ctx(h, 0)), device
        self.ctx.close = lambda: None

    def slot_context(self, k):
        c = Context.__new__(Context)
        c.handle, c.device = C.c_void_p(lib().tcmi_pipeline_ctx(self.handle, k)), self.ctx.device
        c.close = lambda: None
        return c

    def set_orfs(self, start, end, is_plus):
        s_, e_ = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(end, np.int64)
        p_ = np.ascontiguousarray(is_plus, np.uint8)
        check(lib().tcmi_pipeline_set_orfs(self.handle, len(s_), ptr(s_), ptr(e_), ptr(p_)))

    def run(self, readsets, L, mincov, include_ambig, host_reads=None, extra=4096, batch=1, pos_stride=0, out=None):
        """-> (list of consensus bytes, int32 status array).  Raises on the first failed item.
        batch > 1: every read set holds `batch` BAMs (Context.upload_batch at `pos_stride`); the outputs
        (and host_reads) are then per BAM, item-major.
        out: a caller-owned uint8 array of >= n * (L + 1 + extra) bytes — the walkers write every consensus into it
        at multiples of that stride and the call returns (out, lengths, status) without building Python objects
        (a long queue otherwise spends its time in page faults of the fresh buffer and in 30 KB copies)."""
        n_items = len(readsets)
        rs = (C.c_void_p * n_items)(*[r.handle for r in readsets])
        n = n_items * int(batch)
        keep, hr = [], None
        if host_reads is not None:
            by_id = {}                                   # the same reads object may back many items: convert it once
            ptrs = []
            for r in host_reads:
                if r is None:
                    ptrs.append(None)
                    continue
                if id(r) not in by_id:
                    st, k = r.as_struct() if isinstance(r, BamFile) else _ffi.as_reads(r)
                    by_id[id(r)] = (st, k, C.pointer(st))
                ptrs.append(by_id[id(r)][2])
            hr = (C.POINTER(_ffi.Reads) * n)(*ptrs)
            keep.append(by_id)
        stride = int(L) + 1 + int(extra)
        own = out is None
        if own:
            out = np.empty(n * stride, np.uint8)
        elif out.dtype != np.uint8 or not out.flags.c_contiguous or out.size < n * stride:
            raise ValueError("out must be a contiguous uint8 array of at least %d bytes" % (n * stride))
        lens = np.zeros(n, np.int64)
        status = np.zeros(n, np.int32)
        rc = lib().tcmi_pipeline_run_batched(self.handle, n_items, rs, int(batch), int(pos_stride), hr, int(L), int(mincov),
                                             int(bool(include_ambig)), ptr(out), stride, ptr(lens), ptr(status))
        self.last_status = status
        check(rc)
        if not own:
            return out, lens, status
        return [out[i * stride:i * stride + int(lens[i])].tobytes() for i in range(n)], status

    def close(self):
        if self.handle:
            lib().tcmi_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Walker:
    """consensus_walk with the GFF rows fixed and no inserts: the arrays are converted once, so a
    call is one ctypes crossing (the GIL is released meanwhile: walks of different BAMs can run
    on several host threads)."""

    def __init__(self, orf_start, orf_end, orf_is_plus, include_ins=True):
        self.os = np.ascontiguousarray(orf_start, np.int64)
        self.oe = np.ascontiguousarray(orf_end, np.int64)
        self.op = np.ascontiguousarray(orf_is_plus, np.uint8)
        self.off = np.zeros(1, np.int64)
        self.include_ins = int(bool(include_ins))
        self._fn = lib().tcmi_consensus_walk

    def __call__(self, plain, alt, flags):
        L = len(plain)
        out = C.create_string_buffer(L + 1)
        n_out, err = C.c_int64(0), C.c_int64(0)
        n = len(self.os)
        ns, ne = np.empty(n, np.int64), np.empty(n, np.int64)
        rc = self._fn(ptr(plain), ptr(alt), ptr(flags), L, n, ptr(self.os), ptr(self.oe), ptr(self.op),
                      0, None, None, b"", ptr(self.off), self.include_ins, C.cast(out, C.c_void_p), L + 1,
                      C.byref(n_out), ptr(ns), ptr(ne), C.byref(err))
        if rc == _ffi.E_KEYERROR:
            raise WalkKeyError(err.value)
        if rc == _ffi.E_ZERODIV:
            raise ZeroDivisionError("division by zero")
        check(rc)
        return out.raw[:n_out.value], ns, ne


# pysam's defaults for AlignmentFile.pileup() (Events.py:66 passes none): SURVEY §8-Q8
DEFAULT_MIN_BASE_QUALITY = 13
DEFAULT_FLAG_FILTER = 0x4 | 0x100 | 0x200 | 0x400
DEFAULT_MAX_DEPTH = 8000


def modal_tokens(reads, positions, min_base_quality=DEFAULT_MIN_BASE_QUALITY, flag_filter=DEFAULT_FLAG_FILTER,
                 ignore_orphans=True, max_depth=DEFAULT_MAX_DEPTH, ignore_overlaps=True):
    """For each 1-based position: (modal upper-cased token or None, n_tokens) under pysam's default pileup arguments
    (Events.py:66).  Raises TcmiError(E_UNSUPPORTED) where overlapping mates meet a deletion on the column (the one case of
    pysam's overlap handling that is not modelled)."""
    positions = np.ascontiguousarray(sorted(int(p) for p in positions), np.int64)
    n = len(positions)
    if n == 0:
        return {}
    r, keep = reads.as_struct() if isinstance(reads, BamFile) else _ffi.as_reads(reads)
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        off = np.zeros(n + 1, np.int64)
        cnt = np.zeros(n, np.int64)
        deep = C.c_int32(0)
        rc = lib().tcmi_modal_tokens(C.byref(r), n, ptr(positions), int(min_base_quality), int(flag_filter),
                                     int(bool(ignore_orphans)), int(max_depth), int(bool(ignore_overlaps)), C.cast(buf, C.c_void_p), cap,
                                     ptr(off), ptr(cnt), C.byref(deep))
        if rc == _ffi.E_ARG and b"token buffer too small" in (lib().tcmi_last_error(None) or b"") and cap < (1 << 30):
            cap *= 16
            continue
        check(rc)
        break
    del keep
    if deep.value & 2:
        raise _ffi.TcmiError(_ffi.E_UNSUPPORTED, "overlapping mates with a deletion on an insert-candidate column: pysam's overlap "
                                                  "quality tweak there is not modelled")
    out = {}
    for k in range(n):
        tok = buf.raw[off[k]:off[k + 1]].decode("ascii") if cnt[k] else None
        out[int(positions[k])] = (tok, int(cnt[k]))
    return out


class BamFile:
    """A BAM decoded by libtcmi (tcmi_bam_load): pysam.AlignmentFile's role for this path."""

    def __init__(self, path, threads=0):
        h = C.c_void_p()
        check(lib().tcmi_bam_load(str(path).encode(), int(threads), C.byref(h)))
        self.handle = h
        self.filename = str(path)
        n_ref, name, ln = C.c_int32(0), C.c_char_p(), C.c_int64(0)
        check(lib().tcmi_bam_header(h, C.byref(n_ref), C.byref(name), C.byref(ln)))
        self.nreferences = n_ref.value
        self.references = ((name.value or b"").decode(),) if n_ref.value else ()
        self.lengths = (ln.value,) if n_ref.value else ()
        v = [C.c_int64(0), C.c_int32(0)] + [C.c_int64(0) for _ in range(5)]
        check(lib().tcmi_bam_info(h, *[C.byref(x) for x in v]))
        (self.n_reads, self.sorted, self.file_bytes, self.inflated_bytes, self.n_blocks, self.n_cigar,
         self.n_qual) = (x.value for x in v)
        self.text = (lib().tcmi_bam_text(h) or b"").decode("utf-8", "replace")

    def as_struct(self):
        r = _ffi.Reads()
        check(lib().tcmi_bam_reads(self.handle, C.byref(r)))
        return r, self

    def arrays(self):
        """numpy views (owned by this object) in the tcmi_reads layout."""
        r, _ = self.as_struct()
        n = self.n_reads

        def view(p, cnt):
            if cnt == 0:
                return np.zeros(0, np.ctypeslib.as_array(p, shape=(1,)).dtype)
            return np.ctypeslib.as_array(p, shape=(cnt,))
        cig_off = view(r.cigar_off, n + 1)
        seq_off = view(r.seq_off, n + 1)
        name_off = view(r.name_off, n + 1)
        return {"n_reads": n, "pos": view(r.pos, n), "flag": view(r.flag, n), "l_qseq": view(r.l_qseq, n),
                "next_tid": view(r.next_tid, n), "next_pos": view(r.next_pos, n), "tlen": view(r.tlen, n), "name_off": name_off,
                "names": np.ctypeslib.as_array(C.cast(r.names, C.POINTER(C.c_uint8)), shape=(max(1, int(name_off[n]) if n else 1),)),
                "tid": view(r.tid, n), "cigar_off": cig_off, "cigar": view(r.cigar, max(1, self.n_cigar)),
                "seq_off": seq_off, "seq": view(r.seq, max(1, int(seq_off[n]) if n else 1)),
                "qual": view(r.qual, max(1, self.n_qual)), "_owner": self}

    def close(self):
        if self.handle:
            lib().tcmi_bam_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LazyBam:
    """The decoded BAM for the few callers that need reads on the host (insert tokens, Events.py:47-82): decoded by the
    host reader on first use.  The tally itself never needs it — the device decodes the file."""

    def __init__(self, path, threads=0):
        self.filename, self._threads, self._bam = str(path), threads, None

    def get(self):
        if self._bam is None:
            self._bam = BamFile(self.filename, threads=self._threads)
        return self._bam

    def close(self):
        if self._bam is not None:
            self._bam.close()
            self._bam = None


class DeviceBam:
    """A BAM file headed for the device decoder (tcmi_bamfile): the HOST side only reads the bytes into pinned memory,
    walks the BGZF block headers and parses the BAM header; Context.upload_bamfile() inflates, indexes and packs it with
    HIP kernels.  Raises TcmiError(E_UNSUPPORTED) there when the file needs the host reader (BamFile)."""

    def __init__(self, path):
        h = C.c_void_p()
        check(lib().tcmi_bamfile_read(str(path).encode(), C.byref(h)))
        self.handle = h
        self.filename = str(path)
        fb, ib, nb, ln = (C.c_int64(0) for _ in range(4))
        n_ref, name = C.c_int32(0), C.c_char_p()
        check(lib().tcmi_bamfile_info(h, C.byref(fb), C.byref(ib), C.byref(nb), C.byref(n_ref), C.byref(name), C.byref(ln)))
        self.file_bytes, self.inflated_bytes, self.n_blocks = fb.value, ib.value, nb.value
        self.nreferences = n_ref.value
        self.references = ((name.value or b"").decode(),) if n_ref.value else ()
        self.lengths = (ln.value,) if n_ref.value else ()
        self.text = (lib().tcmi_bamfile_text(h) or b"").decode("utf-8", "replace")

    def to_device(self, ctx):
        """The compressed bytes into HBM, to stay (tcmi_bamfile_to_device): uploads of this file then start from device memory."""
        check(lib().tcmi_bamfile_to_device(ctx.handle, self.handle), ctx.handle)
        return self

    def decode_to_host(self, ctx):
        """(for tests / tools) -> (inflated stream uint8, record offsets uint64), both produced by the device."""
        stream = np.empty(max(1, self.inflated_bytes), np.uint8)
        cap = self.inflated_bytes // 36 + 16
        rec = np.empty(cap, np.uint64)
        n = C.c_int64(0)
        check(lib().tcmi_bamfile_decode_to_host(ctx.handle, self.handle, ptr(stream), stream.size, ptr(rec), cap, C.byref(n)), ctx.handle)
        return stream[:self.inflated_bytes], rec[:n.value].copy()

    def close(self):
        if self.handle:
            lib().tcmi_bamfile_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FileRunner:
    """BAM files -> consensus FASTA text (TrueConsense.py:212-264 for many inputs; BASELINE configs[1] / [3]): the native
    runner tcmi_filerunner (csrc/pipeline.cpp).

    Stages per BAM, each on its own threads so that consecutive BAMs overlap:
      read    HOST: file bytes into pinned memory, BGZF block table, BAM header (`decoders` files in flight)
      gpu     H2D of the compressed file, HIP: BGZF inflate + record chain + pack, tally + call, records to pinned host
              memory (`gpu_streams` contexts, each with a stream and a device arena of its own: the kernels of one BAM
              fill the gaps of another's)
      walk    HOST: insert tokens if any candidate, sequential consensus walk, FASTA text (`walkers` threads)
    A file the device decoder does not take (records straddling BGZF blocks, long reads) is decoded by the host reader
    (tcmi_bam_load, `decode_threads` threads) and packed from its flat arrays.  `seconds` accumulates each stage's busy time."""

    def __init__(self, ctx, gff_rows, mincov, include_ambig=True, decoders=2, decode_threads=8, walkers=2, gpu_streams=2):
        device = ctx.device if isinstance(ctx, Context) else int(ctx)
        self.mincov, self.amb = int(mincov), bool(include_ambig)
        h = C.c_void_p()
        check(lib().tcmi_filerunner_create(int(device), int(decoders), max(1, int(gpu_streams)), int(walkers), int(decode_threads), C.byref(h)))
        self.handle = h
        self.n_ctx = max(1, int(gpu_streams))
        s_ = np.ascontiguousarray([r["start"] for r in gff_rows], np.int64)
        e_ = np.ascontiguousarray([r["end"] for r in gff_rows], np.int64)
        p_ = np.ascontiguousarray([r.get("strand") == "+" for r in gff_rows], np.uint8)
        check(lib().tcmi_filerunner_set_orfs(h, len(s_), ptr(s_), ptr(e_), ptr(p_)))
        self.device_decode = True       # BGZF inflate + record index on the GPU; files it does not take go to the host reader
        self.seconds = {"decode": 0.0, "upload": 0.0, "step": 0.0, "walk": 0.0}
        self.decoded_on = {"device": 0, "host": 0}
        self._device = device

    @property
    def contexts(self):
        out = []
        for k in range(self.n_ctx):
            c = Context.__new__(Context)
            c.handle, c.device = C.c_void_p(lib().tcmi_filerunner_ctx(self.handle, k)), self._device
            c.close = lambda: None
            out.append(c)
        return out

    def run(self, paths, names=None, ref_len=0, max_inserted=4096):
        """-> list of FASTA texts, in input order."""
        n = len(paths)
        names = names or ["S%d" % i for i in range(n)]
        if n == 0:
            return []
        c_paths = (C.c_char_p * n)(*[str(p).encode() for p in paths])
        c_names = (C.c_char_p * n)(*[str(x).encode() for x in names])
        stride = int(ref_len) + int(max_inserted) + max(len(x) for x in names) + 64
        # (positions beyond ref_len that the reads reach lengthen the consensus: the runner reports a too small stride)
        out = np.empty(n * stride, np.uint8)
        lens = np.zeros(n, np.int64)
        status = np.zeros(n, np.int32)
        sec = (C.c_double * 4)()
        on = (C.c_int64 * 2)()
        rc = lib().tcmi_filerunner_run(self.handle, n, c_paths, c_names, int(ref_len), self.mincov, int(self.amb), int(bool(self.device_decode)),
                                       ptr(out), stride, ptr(lens), ptr(status), sec, on)
        self.last_status = status
        for k, v in zip(("decode", "upload", "step", "walk"), sec):
            self.seconds[k] += v
        self.decoded_on["device"] += on[0]
        self.decoded_on["host"] += on[1]
        check(rc)
        return [out[i * stride:i * stride + int(lens[i])].tobytes().decode("ascii") for i in range(n)]

    def run_resident(self, dbams, names=None, ref_len=0, max_inserted=4096):
        """... of DeviceBam objects read before (and, after DeviceBam.to_device, resident in HBM): no read stage, no PCIe copy of
        the file inside the run.  -> list of FASTA texts, in input order."""
        n = len(dbams)
        names = names or ["S%d" % i for i in range(n)]
        if n == 0:
            return []
        c_files = (C.c_void_p * n)(*[d.handle for d in dbams])
        c_names = (C.c_char_p * n)(*[str(x).encode() for x in names])
        stride = int(ref_len) + int(max_inserted) + max(len(x) for x in names) + 64
        out = np.empty(n * stride, np.uint8)
        lens = np.zeros(n, np.int64)
        status = np.zeros(n, np.int32)
        sec = (C.c_double * 4)()
        on = (C.c_int64 * 2)()
        rc = lib().tcmi_filerunner_run_resident(self.handle, n, c_files, c_names, int(ref_len), self.mincov, int(self.amb), ptr(out), stride,
                                                ptr(lens), ptr(status), sec, on)
        self.last_status = status
        for k, v in zip(("decode", "upload", "step", "walk"), sec):
            self.seconds[k] += v
        self.decoded_on["device"] += on[0]
        self.decoded_on["host"] += on[1]
        check(rc)
        return [out[i * stride:i * stride + int(lens[i])].tobytes().decode("ascii") for i in range(n)]

    def set_outputs(self, ref_id, ref_seq, vcf_head, gff_head, gff_row_columns):
        """What the native VCF / GFF writers need besides a sample's walk (run_files): the reference's first record, the complete
        VCF header text, the GFF header text, and per GFF row [source, type, score, strand, phase, attributes]."""
        flat = [str(c).encode() for row in gff_row_columns for c in row]
        arr = (C.c_char_p * max(1, len(flat)))(*flat)
        check(lib().tcmi_filerunner_set_outputs(self.handle, str(ref_id).encode(), str(ref_seq).encode(), str(vcf_head).encode(),
                                                str(gff_head).encode(), len(gff_row_columns), arr))

    def run_files(self, paths, names, fasta, vcf=None, gff=None, doc=None, ref_len=0):
        """BAM files -> per sample its consensus FASTA and (lists, entries may be None) VCF, corrected GFF, coverage TSV, all written
        by the native runner's walker threads.  Raises what the reference raises (KeyError, ZeroDivisionError) for the first
        sample that fails; last_status has every sample's code."""
        n = len(paths)
        if n == 0:
            return
        def arr(xs):
            if xs is None:
                return None
            return (C.c_char_p * n)(*[None if x is None else str(x).encode() for x in xs])
        status = np.zeros(n, np.int32)
        sec = (C.c_double * 4)()
        on = (C.c_int64 * 2)()
        rc = lib().tcmi_filerunner_run_files(self.handle, n, arr(paths), arr(names), arr(fasta), arr(vcf), arr(gff), arr(doc), int(ref_len),
                                             self.mincov, int(self.amb), int(bool(self.device_decode)), ptr(status), sec, on)
        self.last_status = status
        for k, v in zip(("decode", "upload", "step", "walk"), sec):
            self.seconds[k] += v
        self.decoded_on["device"] += on[0]
        self.decoded_on["host"] += on[1]
        if rc == _ffi.E_KEYERROR:
            raise KeyError((lib().tcmi_last_error(None) or b"").decode("utf-8", "replace"))
        if rc == _ffi.E_ZERODIV:
            raise ZeroDivisionError("division by zero")
        check(rc)

    def close(self):
        if self.handle:
            lib().tcmi_filerunner_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
