#!/bin/bash
# tools/gpu_final_check.sh TAG — the GPU test suite, smoke(), then the default bench line and the driver's shape, into gpurun_out/TAG
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
set -e
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_shape.json 2> gpurun_out/$tag/bench_driver_shape.err
python3 - <<P
import json
for n in ("bench_default", "bench_driver_shape"):
    d = json.loads(open("gpurun_out/$tag/%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(d["value"] / 1e6, 2), "M positions/s", round(d["ms_per_step"], 4), "ms", d["fasta_bit_exact"], d["fasta_all_timed"]["all_equal_the_oracle_chain"],
          "file leg", round(d["file_to_fasta"]["value"] / 1e6, 2), "hard", round(d["hard_bam"]["pipelined"]["ms_per_bam"], 3), "real", round(d["real_bam"]["pipelined"]["ms_per_bam"], 3),
          "roofline", round(d["roofline"]["frac"], 4), round(d["roofline"]["aggregate"]["frac"], 4))
P
