#!/bin/bash
# headline (bench.py default legs off) at GPU_MAX_HW_QUEUES x contexts, turn about: tools/sweep_queues2.sh "4:8 8:8 8:12 4:8 8:8"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
for cfg in ${@:-4:8 8:8 8:12 4:8 8:8}; do
  q=${cfg%%:*}; s=${cfg##*:}
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --gpu-streams $s --no-cpu-baseline --no-cli-batch --no-configs2 --no-hard-bam --no-resident > gpurun_out/q/q${q}s$s.json 2> gpurun_out/q/q${q}s$s.err || { echo fail $cfg; tail -3 gpurun_out/q/q${q}s$s.err; exit 1; }
  echo "hwq $q contexts $s: $(python3 tools/bench_summary.py gpurun_out/q/q${q}s$s.json | sed -n 1p)"
done
