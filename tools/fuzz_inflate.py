#!/usr/bin/env python3
"""tools/fuzz_inflate.py [seconds=60] [seed=1] — random BAM files through the device decoder against zlib, byte for byte, and through the
tally against the oracle's counts: read sets of random size, qualities from random distributions (constant ... uniform: 35 : 1 ... 2 : 1, so
every bgzf_symbols / bgzf_copy variant comes up), zlib level 1 - 9, strategy default / filtered / RLE / Huffman-only / fixed, memLevel 1 - 9,
blocks cut on records or filled to the brim, tiny blocks; the lanes' token scratch cut down at random (`sym_scratch_div`: pass B)."""
import gzip, os, struct, sys, tempfile, time, zlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle
from trueconsense_amd import engine, synthetic as sy
from trueconsense_amd.io import bamwriter

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref, orfs = sy.make_reference(L=6000, cds=[(100, 2500), (3000, 5800)])
L = len(ref)
ctx = engine.Context(0)
tmp = tempfile.mkdtemp(prefix="tcmi_fuzz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p = os.path.join(tmp, "f.bam")
t0, n, kinds = time.time(), 0, {}
try:
    while time.time() - t0 < seconds:
        nr = int(rng.integers(200, 60000))
        reads = sy.make_reads(ref, nr, seed=int(rng.integers(1, 1 << 30)), indel_sites=sy.default_indel_sites(orfs) if rng.random() < 0.3 else None)
        spread = rng.choice([0, 1, 2, 4, 8, 16, 40])
        if spread == 0:
            q = np.full(len(reads["qual"]), 30, np.uint8)
        else:
            w = np.exp(-0.5 * ((np.arange(2, 42) - 36) / float(spread)) ** 2) + (0.004 if rng.random() < 0.5 else 0.0)
            q = rng.choice(np.arange(2, 42, dtype=np.uint8), size=len(reads["qual"]), p=w / w.sum())
        reads["qual"] = q
        strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED])
        level = int(rng.integers(1, 10))
        kw = dict(level=level)
        r = rng.random()
        if r < 0.3: kw.update(split_records=True)
        if r < 0.1: kw.update(block=int(rng.integers(600, 5000)))
        bamwriter.DEFLATE.update(strategy=int(strategy), mem_level=int(rng.integers(1, 10)), flush_every=int(rng.choice([0, 0, 0, 700, 5000])))
        try:
            bamwriter.write_bam(p, reads, "ref", L, **kw)
        finally:
            bamwriter.DEFLATE.update(strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, flush_every=0)
        div = int(rng.choice([1, 1, 1, 4, 16, 64]))
        ctx.set_option("sym_scratch_div", div)
        want = np.frombuffer(gzip.decompress(open(p, "rb").read()), np.uint8)
        d = engine.DeviceBam(p)
        stream, rec = d.decode_to_host(ctx)
        assert len(stream) == len(want) and np.array_equal(stream, want), ("inflate differs", nr, spread, strategy, level, kw, div, np.argwhere(stream != want)[:3] if len(stream) == len(want) else (len(stream), len(want)))
        rs = ctx.upload_bamfile(d)
        got = ctx.step(rs, max(L, rs.max_end, 1), 30, True)[3]
        ro = c_oracle.read_bam(p)
        wc = c_oracle.tally(ro, len(got))
        assert rs.n_reads == ro["n_reads"] and np.array_equal(got, wc), ("counts differ", nr, spread, strategy, level, kw, div)
        rs.free(); d.close()
        ratio = len(want) / max(1, os.path.getsize(p))
        k = "35:1+" if ratio > 20 else "8-20:1" if ratio > 8 else "4-8:1" if ratio > 4 else "<4:1"
        kinds[k] = kinds.get(k, 0) + 1
        n += 1
finally:
    ctx.set_option("sym_scratch_div", 1)
    for f in os.listdir(tmp): os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
print("fuzz_inflate: %d files in %.0f s, every stream = zlib's and every count matrix = the oracle's; by compression ratio: %s" % (n, time.time() - t0, kinds))
