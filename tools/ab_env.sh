#!/bin/bash
# A/B on one box over an environment knob: tools/ab_env.sh VAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  d=gpurun_out/abenv/${var}_${v}_$rep; mkdir -p $d
  env $var=$v true
  export $var=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 60 --warmup 10 --bams 2 --no-cpu-baseline --serial > $d/bench.json 2> $d/err.log
  echo "$var=$v rep$rep: $(grep tally_fast $d/*/*kernel_stats.csv | cut -d, -f2-6)"
done
done
