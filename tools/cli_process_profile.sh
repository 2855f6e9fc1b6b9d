cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, subprocess, time, json
sys.path.insert(0, os.getcwd())
from trueconsense_amd import synthetic as sy
from trueconsense_amd.io import bamwriter
ref, orfs = sy.make_reference()
reads = sy.make_reads(ref, 10000, seed=77, indel_sites=sy.default_indel_sites(orfs))
d = "/tmp/cliprof"; os.makedirs(d, exist_ok=True); os.chdir(d)
bamwriter.write_bam("in.bam", reads, "MN908947.3", len(ref))
open("r.fa","w").write(">MN908947.3\n"+ref+"\n")
head, body = sy.gff_text(orfs, seqid="MN908947.3"); open("g.gff","w").write(head+body)
env = dict(os.environ, PYTHONPATH=os.environ["GRAFT_REPO_ROOT"])
argv = [sys.executable, "-m", "trueconsense_amd.TrueConsense", "-i", "in.bam", "-ref", "r.fa", "-gff", "g.gff", "-cov", "30", "-name", "S", "-o", "o.fa", "-vcf", "o.vcf", "-ogff", "o.gff", "-doc", "o.tsv", "--stats", "s.json"]
for k in range(4):
    t0 = time.perf_counter(); r = subprocess.run(argv, env=env, capture_output=True, text=True); dt = time.perf_counter() - t0
    print("run %d: %.3f s rc %d" % (k, dt, r.returncode), json.load(open("s.json"))["seconds"])
r = subprocess.run([sys.executable, "-X", "importtime"] + argv[1:], env=env, capture_output=True, text=True)
rows = []
for ln in r.stderr.splitlines():
    if ln.startswith("import time:") and "|" in ln:
        p = ln.split("|")
        try: rows.append((int(p[1]), p[2].strip()))
        except ValueError: pass
rows.sort(reverse=True)
print("top cumulative imports (us):", rows[:14])
r = subprocess.run([sys.executable, "-c", "import time; t=time.perf_counter(); import numpy; print('numpy %.3f' % (time.perf_counter()-t)); t=time.perf_counter(); import ctypes; from trueconsense_amd import _ffi; _ffi.lib(); print('lib load %.3f' % (time.perf_counter()-t)); t=time.perf_counter(); from trueconsense_amd.engine import Context; c=Context(0); print('ctx create %.3f' % (time.perf_counter()-t))"], env=env, capture_output=True, text=True)
print(r.stdout, r.stderr[-300:])
PY
