# Would larger grids pay in the overlapped pipeline?  C contexts (a thread each), every one decoding + packing + tallying + calling
# files of m million reads from HBM-resident bytes, back to back: million reads per second over all contexts.
#   python3 tools/group_proxy.py "8x1 4x2 3x4 2x4 2x8"
import os, sys, time, tempfile, threading, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from trueconsense_amd import synthetic as sy, engine
from trueconsense_amd.io import bamwriter
ref, orfs = sy.make_reference(); L = len(ref)
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths = {}
def file_of(m):
    if m not in paths:
        p = os.path.join(tmp, "g%d.bam" % m)
        n = 1_000_000
        for r in range(m):
            reads = sy.make_reads(ref, n, seed=7000 + r)
            bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6, part=(r == 0, r == m - 1), first_id=r * n)
        paths[m] = p
    return paths[m]
for cfg in (sys.argv[1] if len(sys.argv) > 1 else "8x1 3x4").split():
    TOKENS = cfg.endswith("t")
    C, m = (int(x) for x in cfg.rstrip("t").split("x"))
    p = file_of(m)
    ctxs = [engine.Context(0) for _ in range(C)]
    dbs = [engine.DeviceBam(p).to_device(c) for c in ctxs]
    done = [0] * C
    stop = [False]
    def work(k):
        c, d = ctxs[k], dbs[k]
        while not stop[0]:
            rs, plain, alt, flags, _ = c.bamfile_step(d, L, 30, True, want_counts=False)
            if TOKENS:                                       # as the file runner does: the insert candidates' tokens from the resident stream
                cand = (np.flatnonzero(flags & 8) + 1).tolist()
                if cand: c.readset_modal_tokens(rs, cand)
            rs.free()
            done[k] += 1
    for k in range(C):                                       # warm-up: allocations
        rs, *_ = ctxs[k].bamfile_step(dbs[k], L, 30, True, want_counts=False); rs.free()
    th = [threading.Thread(target=work, args=(k,)) for k in range(C)]
    t0 = time.time()
    for t in th: t.start()
    time.sleep(3.0)
    stop[0] = True
    for t in th: t.join()
    dt = time.time() - t0
    print("%d contexts x %d M reads per call%s: %.1f M reads/s = %.3f ms per million reads" % (C, m, " + insert tokens" if TOKENS else "", sum(done) * m / dt, 1e3 * dt / (sum(done) * m)), flush=True)
    for d in dbs: d.close()
    for c in ctxs: c.close()
