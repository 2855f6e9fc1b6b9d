"""ctypes loader for oracle/tally_oracle.c (TEST INFRASTRUCTURE ONLY, see its header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtcoracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("tally_oracle.c", "bam_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_tally.restype = C.c_int64
        _lib.orc_extent.restype = C.c_int64
        _lib.orc_call.restype = None
        _lib.orc_bam_load.restype = C.c_int
        _lib.orc_bam_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        _lib.orc_bam_dims.restype = None
        _lib.orc_bam_dims.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        _lib.orc_bam_ref0.restype = C.c_char_p
        _lib.orc_bam_ref0.argtypes = [C.c_void_p]
        _lib.orc_bam_fill.restype = None
        _lib.orc_bam_fill.argtypes = [C.c_void_p] * 12
        _lib.orc_bam_free.restype = None
        _lib.orc_bam_free.argtypes = [C.c_void_p]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def extent(reads, ref_len):
    r = reads
    return int(lib().orc_extent(C.c_int64(int(r["n_reads"])), _p(r["pos"], C.c_int32), _p(r["flag"], C.c_uint16),
                                _p(r["cigar_off"], C.c_uint64), _p(r["cigar"], C.c_uint32),
                                _p(r.get("tid"), C.c_int32), C.c_int64(ref_len)))


def tally(reads, L):
    """-> int32 [L,7] in the reference's column order (indexing.py:134)."""
    r = reads
    counts = np.zeros((L, 7), np.int32)
    lib().orc_tally(C.c_int64(int(r["n_reads"])), _p(r["pos"], C.c_int32), _p(r["flag"], C.c_uint16),
                    _p(r["l_qseq"], C.c_int32), _p(r["cigar_off"], C.c_uint64), _p(r["cigar"], C.c_uint32),
                    _p(r["seq_off"], C.c_uint64), _p(r["seq"], C.c_uint8), _p(r.get("tid"), C.c_int32),
                    C.c_int64(L), _p(counts, C.c_int32))
    return counts


def call(counts, mincov, include_ambig):
    counts = np.ascontiguousarray(counts, np.int32)
    L = len(counts)
    plain, alt, flags = (np.empty(L, np.uint8) for _ in range(3))
    lib().orc_call(_p(counts, C.c_int32), C.c_int64(L), C.c_int32(mincov), C.c_int(bool(include_ambig)),
                   _p(plain, C.c_uint8), _p(alt, C.c_uint8), _p(flags, C.c_uint8))
    return plain, alt, flags


def read_bam(path):
    """BAM file -> dict of flat arrays in the tcmi_reads layout (oracle/bam_oracle.c: sequential gunzip of the
    multi-member stream + one record walk, one thread), plus 'ref0_name', 'ref0_len', 'n_ref', 'inflated_bytes'."""
    h = C.c_void_p()
    rc = lib().orc_bam_load(str(path).encode(), C.byref(h))
    if rc:
        raise ValueError("bam_oracle: cannot decode %s (code %d)" % (path, rc))
    try:
        n, nc, ns, nq, lr, infl = (C.c_int64(0) for _ in range(6))
        nref = C.c_int32(0)
        lib().orc_bam_dims(h, C.byref(n), C.byref(nc), C.byref(ns), C.byref(nq), C.byref(nref), C.byref(lr), C.byref(infl))
        n_ = n.value
        r = {"n_reads": n_, "pos": np.empty(n_, np.int32), "flag": np.empty(n_, np.uint16), "l_qseq": np.empty(n_, np.int32),
             "tid": np.empty(n_, np.int32), "mapq": np.empty(n_, np.uint8), "cigar_off": np.empty(n_ + 1, np.uint64),
             "cigar": np.empty(max(1, nc.value), np.uint32), "seq_off": np.empty(n_ + 1, np.uint64),
             "seq": np.empty(max(1, ns.value), np.uint8), "qual_off": np.empty(n_ + 1, np.uint64),
             "qual": np.empty(max(1, nq.value), np.uint8)}
        lib().orc_bam_fill(h, *[r[k].ctypes.data_as(C.c_void_p) for k in
                                ("pos", "flag", "l_qseq", "tid", "mapq", "cigar_off", "cigar", "seq_off", "seq", "qual_off", "qual")])
        r["cigar"], r["seq"], r["qual"] = r["cigar"][:nc.value], r["seq"][:ns.value], r["qual"][:nq.value]
        r.update(ref0_name=(lib().orc_bam_ref0(h) or b"").decode(), ref0_len=lr.value, n_ref=nref.value,
                 inflated_bytes=infl.value)
        return r
    finally:
        lib().orc_bam_free(h)
