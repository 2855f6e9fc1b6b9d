# kernel times of the device BGZF decoder (bgzf_symbols + bgzf_copy, or bgzf_inflate under TCMI_INFLATE_LEGACY=1), from HIP
# events, single stream: python3 tools/inflate_time.py [headline|hard|real] [n_reads]
# (real: distinct names and qualities drawn like an Illumina run's — 3.3 : 1, what samtools writes for real data)
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from trueconsense_amd import synthetic as sy, engine, _ffi
from trueconsense_amd.io import bamwriter
kind = sys.argv[1] if len(sys.argv) > 1 else "headline"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ref, orfs = sy.make_reference(); L = len(ref)
reads = sy.make_reads(ref, n, seed=1)
d = tempfile.mkdtemp(dir="/dev/shm"); p = os.path.join(d, "s.bam")
if kind == "hard":
    rng = np.random.default_rng(1)
    reads["qual"] = rng.choice(np.array([2, 12, 23, 37], np.uint8), size=len(reads["qual"]), p=[0.02, 0.05, 0.13, 0.80])
    names = [b"A00123:45:HXXXXX:%d:%d:%d:%d" % (a, b, c, e) for a, b, c, e in zip(rng.integers(1, 5, n), rng.integers(1101, 2679, n), rng.integers(1000, 33000, n), rng.integers(1000, 37000, n))]
    reads["name_off"] = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.uint64)
    reads["names"] = np.frombuffer(b"".join(names), np.uint8).copy()
    bamwriter.write_bam(p, reads, "MN908947.3", L, level=6)
elif kind == "real":
    rng = np.random.default_rng(2)
    q = rng.choice(np.arange(2, 42, dtype=np.uint8), size=(n, 150), p=(lambda w: w / w.sum())(np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004))
    names = rng.integers(48, 58, (n, 27)).astype(np.uint8)
    names[:, :10] = np.frombuffer(b"A00123:45:", np.uint8)
    bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6, qual=q, names=names)
else:
    bamwriter.write_bam_fast(p, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6)
ctx = engine.Context(0); db = engine.DeviceBam(p)
print(kind, "file MB", db.file_bytes / 1e6, "inflated MB", db.inflated_bytes / 1e6, "blocks", db.n_blocks)
for k in range(8):
    if k == 2:
        ctx.profile(True)                   # (the first launches carry one-time costs)
    try:
        rs = ctx.upload_bamfile(db); rs.free()
    except Exception as e:                  # (early-stop timing builds end in an error on purpose)
        print("upload failed:", e)
for name, kid in (("inflate(symbols)", _ffi.K_INFLATE), ("inflate(copy)", _ffi.K_INFLATE_COPY), ("crc", _ffi.K_CRC), ("records", _ffi.K_RECORDS)):
    ms, k = ctx.profile_get(kid)
    if k: print(name, "us", round(1e3 * ms / k, 1), "launches", k)
t = time.time(); b = engine.BamFile(p, threads=16)
ctx.set_option("device_pack", 0)
rs2 = ctx.upload(b); want = ctx.step(rs2, L, 30, True)[3].copy()
rs = ctx.upload_bamfile(db); got = ctx.step(rs, L, 30, True)[3]
print("counts equal the host reader's:", bool(np.array_equal(got, want)))
os.remove(p)
