"""ctypes loader for oracle/tally_oracle.c (TEST INFRASTRUCTURE ONLY, see its header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtcoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "tally_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
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
