#!/bin/bash
# the bench's headline (other legs off) with trueconsense_amd/lib/var/libbase.so and with the default build, turn about: tools/gpu_bench_lib_ab.sh [reps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab
for r in $(seq ${1:-3}); do for v in base new; do
  lib=$PWD/trueconsense_amd/lib/libtcmi.so; [ $v = base ] && lib=$PWD/trueconsense_amd/lib/var/libbase.so
  TCMI_LIB=$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-batch --no-configs2 --no-hard-bam --no-resident > gpurun_out/lab/$v.json 2> gpurun_out/lab/$v.err || { echo fail; tail -3 gpurun_out/lab/$v.err; exit 1; }
  echo "$v: $(python3 tools/bench_summary.py gpurun_out/lab/$v.json 2>/dev/null | sed -n '1,3p' | tr '\n' ' ')"
done; done
