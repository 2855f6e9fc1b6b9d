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
print("blocks", nb, "(s_memtime ticks = shader cycles)")
own = s[s[:, 3] > 0]                  # every block: its own wavefront's header phases
for k, nm in enumerate(["stage payload", "header + code lengths", "tables"]):
    d = own[:, k + 1] - own[:, k]
    print("  symbols %-22s mean %8.1f  p50 %8.1f  max %8d" % (nm, d.mean(), np.median(d), d.max()))
grp = s[s[:, 6] > 0]                  # the first block of every workgroup: the decoding wavefront's phases (for all its blocks)
for k, nm in ((3, "pass A (per workgroup)"), (4, "chain + scan"), (5, "tokens to their places")):
    d = grp[:, k + 1] - grp[:, k]
    print("  symbols %-22s mean %8.1f  p50 %8.1f  max %8d" % (nm, d.mean(), np.median(d), d.max()))
print("  symbols workgroup total     mean %8.1f" % (grp[:, 6] - grp[:, 0]).mean(), " rounds A %.1f" % grp[:, 8].mean())
print("  symbols kernel span (first start .. last end) %d ticks" % (s[:, 6].max() - s[:, 0].min()))
c = c[c[:, 7] > 0]
print("  copy total mean %.1f max %d | tokens %.0f matches %.0f (not plain: %.0f) rounds %.0f team rounds %.0f (matches in them %.0f)" %
      ((c[:, 1] - c[:, 0]).mean(), (c[:, 1] - c[:, 0]).max(), c[:, 7].mean(), c[:, 5].mean(), c[:, 4].mean(), c[:, 6].mean(), c[:, 8].mean(), c[:, 9].mean()))
if c[:, 10:15].any():                   # (a build with EXTRA=-DTCMI_COPY_PHASES)
    for k, nm in enumerate(["token fetch + batch set-up", "stretch set-up, literals", "plain-match loop (TCMI_LM_ASM)", "other matches (copy_any)", "housekeeping"]):
        print("  copy %-32s mean %9.1f" % (nm, c[:, 10 + k].mean()))
    if os.environ.get("TCMI_COPY_PHASES") == "2":      # (built with EXTRA=-DTCMI_COPY_PHASES=2: the batch set-up in four parts)
        print("  copy   of the set-up: wait for the tokens %.0f, length scan %.0f, classification + far parking %.0f, operands + prefetch %.0f" % (c[:, 2].mean(), c[:, 3].mean(), c[:, 15].mean(), c[:, 10].mean()))
        sys.exit(0)
    h = [c[:, 2] & 0xFFFFFFFF, c[:, 2] >> 32, c[:, 3] & 0xFFFFFFFF, c[:, 3] >> 32, c[:, 15] & 0xFFFFFFFF, c[:, 15] >> 32]
    print("  copy plain matches per block by length: <8: %.0f  8-64: %.0f  65-128: %.0f  129-192: %.0f  193-256: %.0f  257+: %.0f" % tuple(x.mean() for x in h))
