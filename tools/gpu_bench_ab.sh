#!/bin/bash
# file -> FASTA throughput with more reader threads / GPU contexts (and bgzf_symbols at 2 / 1 blocks per workgroup)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abx
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --no-hard-bam "$@" > gpurun_out/abx/$name.json 2> gpurun_out/abx/$name.err || { echo "$name failed"; tail -5 gpurun_out/abx/$name.err; return 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/abx/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), {k:round(x*1e3,2) for k,x in d['e2e_stage_busy_seconds_per_bam'].items()}, d['fasta_bit_exact'])
"
}
run d3g3 A=1 -- || exit 1
run d3g4 A=1 -- --gpu-streams 4 || exit 1
run d4g4 A=1 -- --decoders 4 --gpu-streams 4 || exit 1
run d4g6 A=1 -- --decoders 4 --gpu-streams 6 || exit 1
run d6g6 A=1 -- --decoders 6 --gpu-streams 6 || exit 1
run d6g8 A=1 -- --decoders 6 --gpu-streams 8 || exit 1
