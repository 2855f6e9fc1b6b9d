"""ctypes binding of libtcmi.so (C ABI: include/tcmi.h).

There is no Python or CPU fallback anywhere in this package: if the library is missing or
no gfx950 device is usable, the call raises.  Build it with `python __graft_entry__.py` or
`make -C trueconsense_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCMI_LIB") or os.path.join(_HERE, "lib", "libtcmi.so")   # TCMI_LIB: A/B builds

TCMI_OK = 0
E_NODEVICE, E_HIP, E_ARG, E_NOMEM, E_FORMAT, E_IO, E_KEYERROR, E_ZERODIV, E_UNSUPPORTED = range(-1, -10, -1)
COLS = ("coverage", "A", "T", "C", "G", "X", "I")        # indexing.py:134
F_LOWCOV, F_PRIMX, F_MINDEL, F_INSCAND, F_COVGT, F_COVZERO, F_AMBIG = 1, 2, 4, 8, 16, 32, 64
K_TALLY, K_CALL, K_ZERO, K_TALLY_GENERAL, K_PACK_CLASSIFY, K_PACK, K_INFLATE, K_RECORDS, K_CRC, K_INFLATE_COPY = range(10)


class TcmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtcmi error %d: %s" % (code, msg))
        self.code = code


class Reads(C.Structure):
    """struct tcmi_reads"""
    _fields_ = [("n_reads", C.c_int64),
                ("pos", C.POINTER(C.c_int32)), ("flag", C.POINTER(C.c_uint16)),
                ("l_qseq", C.POINTER(C.c_int32)),
                ("cigar_off", C.POINTER(C.c_uint64)), ("cigar", C.POINTER(C.c_uint32)),
                ("seq_off", C.POINTER(C.c_uint64)), ("seq", C.POINTER(C.c_uint8)),
                ("qual", C.POINTER(C.c_uint8)), ("tid", C.POINTER(C.c_int32)),
                ("qual_off", C.POINTER(C.c_uint64)), ("sorted_max_span", C.c_int64),
                ("next_tid", C.POINTER(C.c_int32)), ("next_pos", C.POINTER(C.c_int32)), ("tlen", C.POINTER(C.c_int32)),
                ("name_off", C.POINTER(C.c_uint64)), ("names", C.POINTER(C.c_char))]


_P = C.POINTER
_vp, _i32, _i64, _int, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_uint32
_SIGS = {
    "tcmi_abi_version": (_int, []),
    "tcmi_last_error": (C.c_char_p, [_vp]),
    "tcmi_device_count": (_int, [_P(_int)]),
    "tcmi_ctx_create": (_int, [_int, _P(_vp)]),
    "tcmi_ctx_create_on_stream": (_int, [_int, _vp, _P(_vp)]),
    "tcmi_ctx_destroy": (_int, [_vp]),
    "tcmi_ctx_sync": (_int, [_vp]),
    "tcmi_ctx_stream": (_vp, [_vp]),
    "tcmi_ctx_set_option": (_int, [_vp, C.c_char_p, _int]),
    "tcmi_ctx_stat": (_int, [_vp, C.c_char_p, _P(_i64)]),
    "tcmi_profile_enable": (_int, [_vp, _int]),
    "tcmi_profile_reset": (_int, [_vp]),
    "tcmi_profile_get": (_int, [_vp, _int, _P(C.c_double), _P(_i64)]),
    "tcmi_reads_extent": (_int, [_P(Reads), _i64, _P(_i64)]),
    "tcmi_readset_upload": (_int, [_vp, _P(Reads), _P(_vp)]),
    "tcmi_readset_upload_batch": (_int, [_vp, _vp, _i32, _i64, _P(_vp)]),
    "tcmi_readset_free": (_int, [_vp, _vp]),
    "tcmi_readset_info": (_int, [_vp, _P(_i64), _P(_i64), _P(_i64), _P(_i64), _P(_i64)]),
    "tcmi_readset_sets": (_int, [_vp, _P(_i64), _P(_i64), _P(_i64)]),
    "tcmi_readset_origin": (_int, [_vp, _P(_i32)]),
    "tcmi_tally_dev": (_int, [_vp, _vp, _i64, _i64, _vp, _int]),
    "tcmi_tally": (_int, [_vp, _P(Reads), _i64, _vp]),
    "tcmi_counts_download": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "tcmi_counts_upload": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "tcmi_call_dev": (_int, [_vp, _vp, _i64, _i64, _i32, _int, _vp, _vp, _vp, _vp, _vp]),
    "tcmi_call": (_int, [_vp, _vp, _i64, _i32, _int, _vp, _vp, _vp, _vp, _P(_i64)]),
    "tcmi_step": (_int, [_vp, _vp, _i64, _i32, _int, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _P(_i64)]),
    "tcmi_step_begin": (_int, [_vp, _vp, _i64, _i32, _int, _int]),
    "tcmi_step_end": (_int, [_vp, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _P(_i64)]),
    "tcmi_consensus_walk": (_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, C.c_char_p, _vp,
                                   _int, _vp, _i64, _P(_i64), _vp, _vp, _P(_i64)]),
    "tcmi_modal_tokens": (_int, [_P(Reads), _i32, _vp, _i32, _u32, _int, _i64, _int, _vp, _i64, _vp, _vp, _P(_i32)]),
    "tcmi_pipeline_create": (_int, [_int, _int, _int, _P(_vp)]),
    "tcmi_pipeline_destroy": (_int, [_vp]),
    "tcmi_pipeline_set_orfs": (_int, [_vp, _i32, _vp, _vp, _vp]),
    "tcmi_pipeline_ctx": (_vp, [_vp, _int]),
    "tcmi_pipeline_run": (_int, [_vp, _i64, _vp, _vp, _i64, _i32, _int, _vp, _i64, _vp, _vp]),
    "tcmi_pipeline_run_batched": (_int, [_vp, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _int, _vp, _i64, _vp, _vp]),
    "tcmi_bam_load": (_int, [C.c_char_p, _int, _P(_vp)]),
    "tcmi_bam_free": (_int, [_vp]),
    "tcmi_bam_reads": (_int, [_vp, _P(Reads)]),
    "tcmi_bam_header": (_int, [_vp, _P(_i32), _P(C.c_char_p), _P(_i64)]),
    "tcmi_bam_info": (_int, [_vp, _P(_i64), _P(_i32), _P(_i64), _P(_i64), _P(_i64), _P(_i64), _P(_i64)]),
    "tcmi_bam_text": (C.c_char_p, [_vp]),
    "tcmi_bamfile_read": (_int, [C.c_char_p, _P(_vp)]),
    "tcmi_bamfile_free": (_int, [_vp]),
    "tcmi_bamfile_info": (_int, [_vp, _P(_i64), _P(_i64), _P(_i64), _P(_i32), _P(C.c_char_p), _P(_i64)]),
    "tcmi_bamfile_text": (C.c_char_p, [_vp]),
    "tcmi_bamfile_path": (C.c_char_p, [_vp]),
    "tcmi_bamfile_to_device": (_int, [_vp, _vp]),
    "tcmi_readset_from_bamfile": (_int, [_vp, _vp, _P(_vp), _P(_i64)]),
    "tcmi_readset_from_bamfile_blocks": (_int, [_vp, _vp, _i64, _i64, _P(_vp), _P(_i64)]),
    "tcmi_readset_range_anchors": (_int, [_vp, _P(_i64), _P(_i64)]),
    "tcmi_bamfile_step": (_int, [_vp, _vp, _i64, _i32, _int, _P(_vp), _P(_i64), _P(_vp), _P(_vp), _P(_vp), _P(_vp), _P(_i64)]),
    "tcmi_readset_modal_tokens": (_int, [_vp, _vp, _i32, _vp, _i32, _u32, _int, _i64, _int, _vp, _i64, _vp, _vp, _P(_i32)]),
    "tcmi_split_step": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i32, _int, _vp, _vp, _int, _int, _P(_vp), _P(_vp), _P(_vp), _P(_vp)]),
    "tcmi_readset_ins_entries": (_int, [_vp, _vp, _i32, _vp, _u32, _int, _vp, _i64, _vp, _vp, _i64, _P(_i64)]),
    "tcmi_ins_entries_rebase": (_int, [_vp, _i64, _i64]),
    "tcmi_modal_from_entries": (_int, [_i32, _vp, _vp, _i32, _i64, _int, _vp, _i64, _vp, _i64, _vp, _vp, _P(_i32)]),
    "tcmi_bamfile_decode_to_host": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _P(_i64)]),
    "tcmi_filerunner_create": (_int, [_int, _int, _int, _int, _int, _P(_vp)]),
    "tcmi_filerunner_destroy": (_int, [_vp]),
    "tcmi_filerunner_set_orfs": (_int, [_vp, _i32, _vp, _vp, _vp]),
    "tcmi_filerunner_ctx": (_vp, [_vp, _int]),
    "tcmi_filerunner_run": (_int, [_vp, _i64, _vp, _vp, _i64, _i32, _int, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tcmi_filerunner_run_resident": (_int, [_vp, _i64, _vp, _vp, _i64, _i32, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tcmi_filerunner_set_outputs": (_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, _i32, _vp]),
    "tcmi_filerunner_run_files": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _int, _int, _vp, _vp, _vp]),
}

_lib = None


def _preload_hip_runtime():
    """One process must hold ONE HIP/HSA runtime.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so (soname libamdhip64.so.7, same as /opt/rocm's): if libtcmi pulled in
    /opt/rocm's copy first, a later `import torch` would bring up a second runtime that sees no
    GPU.  So when torch is installed its copy is loaded first (without importing torch) and
    libtcmi's NEEDED libamdhip64.so.7 binds to it; without torch, /opt/rocm's is used."""
    if os.environ.get("TCMI_HIP_RUNTIME") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C trueconsense_amd/csrc` (needs hipcc). "
                "trueconsense_amd has no CPU fallback." % LIB_PATH)
        _preload_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.tcmi_abi_version() != 5:
            raise ImportError("libtcmi ABI version %d, expected 5" % handle.tcmi_abi_version())
        _lib = handle
    return _lib


RCCL_LIB_PATH = os.path.join(_HERE, "lib", "libtcmi_rccl.so")
RCCL_ID_BYTES = 128
_RCCL_SIGS = {
    "tcmi_rccl_unique_id": (_int, [_vp]),
    "tcmi_rccl_comm_init": (_int, [_int, _int, _vp, _P(_vp)]),
    "tcmi_rccl_comm_destroy": (_int, [_vp]),
    "tcmi_rccl_comm_info": (_int, [_vp, _P(_int), _P(_int), _P(_int)]),
    "tcmi_rccl_reduce": (_int, [_vp, _vp, _i64, _vp]),
    "tcmi_rccl_last_error": (C.c_char_p, []),
}
_rccl = None


class RcclUser(C.Structure):
    """struct tcmi_rccl_user (include/tcmi_rccl.h): the `user` of tcmi_rccl_reduce."""
    _fields_ = [("comm", _vp), ("root", _int)]


def rccl_lib():
    """libtcmi_rccl.so (include/tcmi_rccl.h): tcmi_split_step's reduce hook over RCCL and the calls that make a communicator for it.
    One process must hold ONE librccl too: torch's wheel bundles its own (soname librccl.so.1, as /opt/rocm's), so torch's copy is
    loaded first when torch is installed and the hook library's NEEDED entry binds to it."""
    global _rccl
    if _rccl is None:
        if not os.path.exists(RCCL_LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C trueconsense_amd/csrc` (needs hipcc and librccl)" % RCCL_LIB_PATH)
        lib()                                                        # (the HIP runtime first)
        if os.environ.get("TCMI_HIP_RUNTIME") != "system":
            try:
                import importlib.util
                spec = importlib.util.find_spec("torch")
            except (ImportError, ValueError):
                spec = None
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so") if spec is not None and spec.origin else ""
            if cand and os.path.exists(cand):
                C.CDLL(cand)                                         # (RTLD_LOCAL: its symbols in the global scope upset torch's own libraries at exit; the soname is what the hook binds by)
        handle = C.CDLL(RCCL_LIB_PATH)
        for name, (res, args) in _RCCL_SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _rccl = handle
    return _rccl


def check(rc, ctx=None):
    if rc != TCMI_OK:
        msg = lib().tcmi_last_error(ctx)
        raise TcmiError(rc, (msg or b"").decode("utf-8", "replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def as_reads(d):
    """dict of flat numpy arrays (tcmi_reads layout) -> (Reads struct, keep-alive list)."""
    keep = {}

    def arr(key, dt, required=True):
        v = d.get(key)
        if v is None:
            if required:
                raise KeyError(key)
            return None
        v = np.ascontiguousarray(v, dtype=dt)
        keep[key] = v
        return v

    r = Reads()
    r.n_reads = int(d["n_reads"])
    for key, dt, ct, req in (("pos", np.int32, C.c_int32, True), ("flag", np.uint16, C.c_uint16, True),
                             ("l_qseq", np.int32, C.c_int32, True), ("cigar_off", np.uint64, C.c_uint64, True),
                             ("cigar", np.uint32, C.c_uint32, True), ("seq_off", np.uint64, C.c_uint64, True),
                             ("seq", np.uint8, C.c_uint8, True), ("qual", np.uint8, C.c_uint8, False),
                             ("tid", np.int32, C.c_int32, False), ("qual_off", np.uint64, C.c_uint64, False),
                             ("next_tid", np.int32, C.c_int32, False), ("next_pos", np.int32, C.c_int32, False),
                             ("tlen", np.int32, C.c_int32, False), ("name_off", np.uint64, C.c_uint64, False)):
        a = arr(key, dt, req)
        setattr(r, key, a.ctypes.data_as(C.POINTER(ct)) if a is not None else None)
    nm = d.get("names")
    if nm is not None and d.get("name_off") is not None:
        nm = np.ascontiguousarray(np.frombuffer(nm, np.uint8) if isinstance(nm, (bytes, bytearray)) else nm, np.uint8)
        if nm.size == 0:
            nm = np.zeros(1, np.uint8)
        keep["names"] = nm
        r.names = nm.ctypes.data_as(C.POINTER(C.c_char))
    r.sorted_max_span = int(d.get("sorted_max_span", 0) or 0)
    return r, keep


def device_count():
    n = _int(0)
    check(lib().tcmi_device_count(C.byref(n)))
    return n.value
