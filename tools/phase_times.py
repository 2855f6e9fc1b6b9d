#!/usr/bin/env python3
"""tools/phase_times.py [BATCH] — where a chunk workgroup of tally_planes_kernel spends its time.
Needs a diagnostic build (tools/build_variant.sh dbg0 -DTCMI_ABL=256 [-DTCMI_DBG_WAVE=k]) selected with TCMI_LIB=...:
lane 0 of one wave of every chunk workgroup stamps the 100 MHz wall clock at its phase boundaries."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trueconsense_amd import _ffi, synthetic as sy          # noqa: E402
from trueconsense_amd.engine import Context                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ref, orfs = sy.make_reference()
L, stride = len(ref), 29952
from concurrent.futures import ThreadPoolExecutor           # noqa: E402
with ThreadPoolExecutor(8) as ex:
    groups = [list(ex.map(lambda k: sy.make_reads(ref, 1_000_000, seed=77 + 10 * g + k), range(B))) for g in range(2)]
SLOTS = 64
with Context(0) as ctx:
    rss = [ctx.upload_batch(g, stride) if B > 1 else ctx.upload(g[0]) for g in groups]
    a, c, g_ = (C.c_int64(0) for _ in range(3))
    _ffi.check(_ffi.lib().tcmi_readset_sets(rss[0].handle, C.byref(a), C.byref(c), C.byref(g_)))
    n_chunks = min(int(c.value), 8192)
    for rep in range(6):                                      # warm, alternate the two read sets (more than the Infinity Cache)
        ctx.step(rss[rep % 2], B * stride if B > 1 else L, 30, True, want_counts=False)
    out = np.zeros(n_chunks * SLOTS, np.uint64)
    fn = _ffi.lib().tcmi_debug_phase_times
    fn.argtypes = [C.c_void_p, C.c_int]
    fn.restype = C.c_int
    assert fn(out.ctypes.data_as(C.c_void_p), n_chunks) == 0
t = out.reshape(n_chunks, SLOTS).astype(np.int64)
n_stamps = (t != 0).sum(1)
print("chunks", n_chunks, "stamps per chunk: min %d median %d max %d" % (n_stamps.min(), np.median(n_stamps), n_stamps.max()))
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0                              # us
print("workgroup start times (us): min %.1f  25%% %.1f  50%% %.1f  75%% %.1f  max %.1f" % tuple(np.percentile(start, [0, 25, 50, 75, 100])))
sel = n_stamps == np.bincount(n_stamps).argmax()            # the common shape
ts = t[sel]
k = int(np.bincount(n_stamps).argmax())
n_stage = (k - 1 - 6) // 6
print("common shape: %d stamps = %d stages; %d workgroups" % (k, n_stage, sel.sum()))
d = np.diff(ts[:, :k], axis=1) / 100.0
names = ["prologue->stage0 top"]
for s_ in range(n_stage):
    names += ["s%d wait loads" % s_, "s%d headers+stores+issue" % s_, "s%d barrier1" % s_, "s%d coverage" % s_, "s%d inner loop" % s_,
              "s%d barrier2->next top" % s_]
names += ["spread+stores", "barrier", "sum slices", "coverage scan", "atomics"]
tot = (ts[:, k - 1] - ts[:, 0]) / 100.0
print("workgroup lifetime (us): mean %.2f  median %.2f  p90 %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 90)))
agg = {}
for i, nm in enumerate(names[:d.shape[1]]):
    key = nm.split(" ", 1)[1] if nm[0] == "s" and nm[1].isdigit() else nm
    agg.setdefault(key, []).append(d[:, i].mean())
    print("  %-28s mean %.2f us" % (nm, d[:, i].mean()))
print("summed over the stages:")
for key, v in agg.items():
    print("  %-28s %.2f us  (%.0f %%)" % (key, sum(v), 100 * sum(v) / tot.mean()))
end = (t[np.arange(n_chunks), n_stamps - 1] - t0) / 100.0
print("last workgroup ends at %.1f us after the first starts" % end.max())
