#!/bin/bash
# A/B on one box: for each variant, rocprofv3 kernel stats of a short serial bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  d=gpurun_out/ab/${v}_$rep; mkdir -p $d
  TCMI_LIB=$GRAFT_REPO_ROOT/trueconsense_amd/lib/variants/libtcmi_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 60 --warmup 10 --bams 2 --no-cpu-baseline --serial > $d/bench.json 2> $d/err.log
  echo "$v rep$rep: $(grep tally_fast $d/*/*kernel_stats.csv | cut -d, -f2-6)"
done
done
