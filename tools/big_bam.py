# One BAM file whose inflated stream is larger than 4 GiB (>= 16 M reads x 273 bytes), decoded, indexed, packed and tallied on ONE
# GPU in one go: every offset of the device decoder that could have been 32 bits wide is exercised.  Counts against the scalar
# C oracle (the checker; outside any clock).   python3 tools/big_bam.py [million_reads=16]
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import c_oracle
from trueconsense_amd import synthetic as sy, engine
from trueconsense_amd.io import bamwriter
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ref, orfs = sy.make_reference(); L = len(ref)
tile = L - 150 + 1
path = os.path.join(tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None), "big.bam")
want = np.zeros((L, 7), np.int64)
t0 = time.time()
n = 1_000_000
for r in range(m):
    reads = sy.make_reads(ref, n, seed=9000 + r, start_range=(tile * r // m, tile * (r + 1) // m))
    bamwriter.write_bam_fast(path, reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", L, level=6, part=(r == 0, r == m - 1), first_id=r * n)
    want += c_oracle.tally(reads, L)
    if r % 4 == 3: print("written", r + 1, "M reads, %.0f s" % (time.time() - t0), flush=True)
ctx = engine.Context(0)
d = engine.DeviceBam(path)
print("file MB %.1f inflated GiB %.3f blocks %d" % (d.file_bytes / 1e6, d.inflated_bytes / 2**30, d.n_blocks), flush=True)
t1 = time.time()
rs = ctx.upload_bamfile(d)
got = ctx.step(rs, L, 30, True)[3]
print("reads %d piled %d decode + pack + tally %.1f ms (first call: allocations included)" % (rs.n_reads, rs.n_piled, 1e3 * (time.time() - t1)))
print("counts equal the oracle's:", bool(np.array_equal(got, want)), "coverage sum", int(got[:, 0].sum()), "expected", 150 * n * m)
print("decoded in batches of blocks (token scratch %d MiB):" % 4096, ctx.stat("decode_batched") > 0, "| one-sync path taken:", ctx.stat("one_sync_taken"))
rs.free()
ctx.profile(True)
t1 = time.time()
rs = ctx.upload_bamfile(d)
ctx.step(rs, L, 30, True, want_counts=False)
print("second call: decode + pack + tally %.1f ms" % (1e3 * (time.time() - t1)), {k: round(ctx.profile_get(v)[0] * 1e3) for k, v in
      (("symbols_us", engine._ffi.K_INFLATE), ("copy_us", engine._ffi.K_INFLATE_COPY), ("crc_us", engine._ffi.K_CRC), ("classify_us", engine._ffi.K_PACK_CLASSIFY),
       ("pack_us", engine._ffi.K_PACK), ("tally_us", engine._ffi.K_TALLY))})
ctx.profile(False)
# ... and the same file as two block ranges, each decoded on its own (what two ranks would do)
rs.free()
acc = np.zeros_like(got)
half = d.n_blocks // 2
for first, count in ((0, half), (half, d.n_blocks - half)):
    rs = ctx.upload_bamfile(d, blocks=(first, count))
    acc += ctx.step(rs, L, 30, True)[3]
    rs.free()
anchors = []
for first, count in ((0, half), (half, d.n_blocks - half)):
    rs = ctx.upload_bamfile(d, blocks=(first, count))
    anchors.append((first, count) + rs.range_anchors)
    rs.free()
from trueconsense_amd.distributed import check_range_anchors
print("two block ranges add up to the same:", bool(np.array_equal(acc, want)))
print("the ranges' anchors join:", check_range_anchors(anchors, d.inflated_bytes) is None, anchors)
os.remove(path)
