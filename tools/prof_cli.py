import os, sys, tempfile, cProfile, pstats, io
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from trueconsense_amd import synthetic as sy
from trueconsense_amd.io import bamwriter
n = 1000000
d = tempfile.mkdtemp(prefix="tcmi_e2e_")
ref, orfs = sy.make_reference()
reads = sy.make_reads(ref, n, seed=5)
bamwriter.write_bam_fast(os.path.join(d, "in.bam"), reads["pos"], reads["flag"], reads["seq"].reshape(n, -1), 150, "MN908947.3", len(ref))
open(os.path.join(d, "ref.fa"), "w").write(">MN908947.3 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
head, body = sy.gff_text(orfs)
open(os.path.join(d, "f.gff"), "w").write(head + body)
from trueconsense_amd import TrueConsense as cli
argv = ["-i", os.path.join(d, "in.bam"), "-ref", os.path.join(d, "ref.fa"), "-gff", os.path.join(d, "f.gff"), "-cov", "30", "-name", "S",
        "-o", os.path.join(d, "o.fa"), "-vcf", os.path.join(d, "o.vcf"), "-ogff", os.path.join(d, "o.gff"), "-doc", os.path.join(d, "o.tsv")]
cli.main(argv); cli.main(argv)
pr = cProfile.Profile(); pr.enable(); cli.main(argv); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
