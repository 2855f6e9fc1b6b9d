#!/bin/bash
# the bench's headline with a context option at two values, turn about: tools/gpu_ctxopt_ab.sh KEY A B [reps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cab
key=$1; a=$2; b=$3
for r in $(seq ${4:-3}); do for v in $a $b; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-batch --no-configs2 --no-hard-bam --no-resident --ctx-option $key=$v > gpurun_out/cab/$v.json 2> gpurun_out/cab/$v.err || { echo fail; tail -3 gpurun_out/cab/$v.err; exit 1; }
  echo "$key=$v: $(python3 tools/bench_summary.py gpurun_out/cab/$v.json 2>/dev/null | sed -n '1,3p' | tr '\n' ' ')"
done; done
