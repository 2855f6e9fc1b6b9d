#!/bin/bash
# the GPU-side bound (tools/group_proxy.py: contexts in Python threads, no walk, no FASTA) next to the bench's headline on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pvb
timeout -k 10 300 python3 tools/group_proxy.py "8x1t 12x1t" 2>&1 | grep -v amdgpu.ids
for cfg in "8 2" "12 2" "8 3" "12 3" "8 2"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --gpu-streams $1 --walkers $2 --no-cpu-baseline --no-cli-batch --no-configs2 --no-hard-bam --no-resident > gpurun_out/pvb/s$1w$2.json 2> gpurun_out/pvb/s$1w$2.err || { echo fail; tail -3 gpurun_out/pvb/s$1w$2.err; exit 1; }
  echo "bench contexts $1 walkers $2: $(python3 tools/bench_summary.py gpurun_out/pvb/s$1w$2.json 2>/dev/null | sed -n '1,2p' | tr '\n' ' ')"
done
