# kernel time of bgzf_inflate alone for a lib variant: python3 kt.py
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from trueconsense_amd import synthetic as sy, engine, _ffi
from trueconsense_amd.io import bamwriter
ref, orfs = sy.make_reference(); L = len(ref)
n = 1000000
reads = sy.make_reads(ref, n, seed=1)
d = tempfile.mkdtemp(dir="/dev/shm"); p = os.path.join(d, "s.bam")
bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)
ctx = engine.Context(0); db = engine.DeviceBam(p)
for k in range(8):
    if k == 2:
        ctx.profile(True)                   # (the first launches carry one-time costs)
    try:
        rs = ctx.upload_bamfile(db); rs.free()
    except Exception as e:                  # (early-stop timing builds end in an error on purpose)
        pass
ms, k = ctx.profile_get(_ffi.K_INFLATE); print("inflate us", 1e3 * ms / k, k)
