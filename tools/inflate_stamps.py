# phase clocks of bgzf_symbols / bgzf_copy (TCMI_INFLATE_STAMPS diagnostic): python3 tools/inflate_stamps.py [headline|hard] [n_reads]
import os, sys, tempfile, subprocess, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
kind = sys.argv[1] if len(sys.argv) > 1 else "headline"
n = sys.argv[2] if len(sys.argv) > 2 else "1000000"
path = os.path.join(tempfile.mkdtemp(dir="/dev/shm"), "stamps.bin")
env = dict(os.environ, TCMI_INFLATE_STAMPS=path)
subprocess.run([sys.executable, os.path.join(root, "tools/inflate_time.py"), kind, n], env=env, check=True)
a = np.fromfile(path, np.uint64).astype(np.int64)
nb = a.size // 32
s, c = a[:nb * 16].reshape(nb, 16), a[nb * 16:].reshape(nb, 16)
ok = s[:, 6] > 0
s = s[ok]
print("blocks", nb, "with symbols", ok.sum(), "(s_memtime ticks; 100 MHz constant clock => 10 ns each)")
names = ["stage payload", "header + code lengths", "tables", "pass A", "chain + scan", "pass B"]
for k, nm in enumerate(names):
    d = s[:, k + 1] - s[:, k]
    print("  symbols %-22s mean %8.1f  p50 %8.1f  max %8d" % (nm, d.mean(), np.median(d), d.max()))
print("  symbols total               mean %8.1f" % (s[:, 6] - s[:, 0]).mean(), " rounds A %.1f B %.1f" % (s[:, 8].mean(), s[:, 9].mean()))
print("  symbols kernel span (first start .. last end) %d ticks" % (s[:, 6].max() - s[:, 0].min()))
c = c[c[:, 7] > 0]
print("  copy total mean %.1f  prep %.1f  matches %.1f  housekeeping %.1f | tokens %.0f matches %.0f rounds %.0f" %
      ((c[:, 1] - c[:, 0]).mean(), c[:, 2].mean(), c[:, 3].mean(), c[:, 4].mean(), c[:, 7].mean(), c[:, 5].mean(), c[:, 6].mean()))
print("  copy kernel span %d ticks" % (c[:, 1].max() - c[:, 0].min()))
