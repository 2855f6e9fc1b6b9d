#!/usr/bin/env python3
"""tools/soak.py [seconds] — the file -> FASTA runner on the benchmark's 8 BAM files, cycled for a while; every FASTA of every
round is compared with the first round's answer for that file (a race or a rare decoder slip would show as a difference)."""
import os, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from trueconsense_amd import synthetic as sy                     # noqa: E402
from trueconsense_amd.engine import Context, DeviceBam, FileRunner   # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ref, orfs = sy.make_reference()
L = len(ref)
tmp = tempfile.mkdtemp(prefix="tcmi_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths, _ = bench.write_inputs(tmp, ref, orfs, 8, 1_000_000, 0, False, 6)
hard, _ = bench.write_inputs(tmp + "_i", ref, orfs, 4, 300_000, 1, True, 6) if os.makedirs(tmp + "_i", exist_ok=True) is None else (None, None)
paths = paths + hard
# ... and two files that compress like real data (2.5 : 1 and 6 : 1: bgzf_symbols<1, windowed> / bgzf_copy<true, true> and <true, false>)
import numpy as np                                               # noqa: E402
from trueconsense_amd.io import bamwriter                        # noqa: E402
rng = np.random.default_rng(7)
nr = 200_000
rr = sy.make_reads(ref, nr, seed=77)
names = rng.integers(48, 58, (nr, 27)).astype(np.uint8)
names[:, :10] = np.frombuffer(b"A00123:45:", np.uint8)
for tag, q in (("real", rng.choice(np.arange(2, 42, dtype=np.uint8), size=(nr, 150), p=(lambda w: w / w.sum())(np.exp(-0.5 * ((np.arange(2, 42) - 36) / 6.0) ** 2) + 0.004))),
               ("hard", rng.choice(np.array([2, 12, 23, 37], np.uint8), size=(nr, 150), p=[0.02, 0.05, 0.13, 0.80]))):
    p = os.path.join(tmp, "soak_%s.bam" % tag)
    bamwriter.write_bam_fast(p, rr["pos"], rr["flag"], rr["seq"].reshape(nr, -1), 150, "MN908947.3", L, level=6, qual=q, names=names)
    paths.append(p)
ctx = Context(0)
runner = FileRunner(ctx, [{"start": o["start"], "end": o["end"], "strand": o["strand"]} for o in orfs], 30, True, decoders=3, decode_threads=8, walkers=2, gpu_streams=8)
want = runner.run(paths, names=["S"] * len(paths), ref_len=L)
dbams = [DeviceBam(p).to_device(ctx) for p in paths]             # ... and the same files with their bytes resident in HBM, turn about
t0, n, bad, turn = time.time(), 0, 0, 0
while time.time() - t0 < seconds:
    turn += 1
    if turn % 2:
        got = runner.run([paths[i % len(paths)] for i in range(96)], names=["S"] * 96, ref_len=L)
    else:
        got = runner.run_resident([dbams[i % len(paths)] for i in range(96)], names=["S"] * 96, ref_len=L)
    for i, g in enumerate(got):
        n += 1
        if g != want[i % len(paths)]:
            bad += 1
    print("files", n, "differences", bad, flush=True)
print("soak:", n, "files,", bad, "differences,", runner.decoded_on)
sys.exit(1 if bad else 0)
