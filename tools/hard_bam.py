# how does the device decoder do on a BAM that compresses like real data? (Illumina-style names, binned random qualities)
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from trueconsense_amd import synthetic as sy, engine, _ffi
from trueconsense_amd.io import bamwriter
ref, orfs = sy.make_reference(); L = len(ref)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
reads = sy.make_reads(ref, n, seed=3)
rng = np.random.default_rng(1)
reads["qual"] = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=len(reads["qual"]), p=[0.02, 0.05, 0.13, 0.80])
names = [("A00123:45:HXXXXX:%d:%d:%d:%d" % (rng.integers(1, 5), rng.integers(1101, 2679), rng.integers(1000, 33000), rng.integers(1000, 37000))).encode() for _ in range(n)]
reads["name_off"] = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.uint64)
reads["names"] = np.frombuffer(b"".join(names), np.uint8).copy()
d = tempfile.mkdtemp(dir="/dev/shm")
p = os.path.join(d, "hard.bam")
bamwriter.write_bam(p, reads, "MN908947.3", L, level=6)
ctx = engine.Context(0)
db = engine.DeviceBam(p)
print("file MB", db.file_bytes / 1e6, "inflated MB", db.inflated_bytes / 1e6, "ratio", db.inflated_bytes / db.file_bytes, "blocks", db.n_blocks)
ctx.profile(True)
for _ in range(4):
    rs = ctx.upload_bamfile(db); rs.free()
ms, k = ctx.profile_get(_ffi.K_INFLATE)
print("inflate ms per file", ms / k, "-> per 1M reads", ms / k * 1e6 / n)
rs = ctx.upload_bamfile(db)
got = ctx.step(rs, L, 30, True)[3].copy()
t = time.time(); b = engine.BamFile(p, threads=16); print("host reader 16 threads ms", (time.time() - t) * 1e3)
# (a measurement tool: the check is the product's other decoder — host reader + host-array upload — not the test oracle)
ctx.set_option("device_pack", 0)
rs2 = ctx.upload(b)
want = ctx.step(rs2, L, 30, True)[3]
print("counts exact", bool(np.array_equal(got, want)))
