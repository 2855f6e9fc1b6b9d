#!/bin/bash
# file -> FASTA throughput with bgzf_symbols at 4 / 2 / 1 blocks per workgroup, and with more reader threads / GPU contexts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abx
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline "$@" > gpurun_out/abx/$name.json 2> gpurun_out/abx/$name.err || { echo "$name failed"; tail -5 gpurun_out/abx/$name.err; return 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/abx/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), {k:round(x*1e3,2) for k,x in d['e2e_stage_busy_seconds_per_bam'].items()}, d['fasta_bit_exact'])
"
}
run nb2 TCMI_SYM_BLOCKS=2 -- || exit 1
run nb4 TCMI_SYM_BLOCKS=4 -- || exit 1
run nb1 TCMI_SYM_BLOCKS=1 -- || exit 1
run legacy TCMI_INFLATE_LEGACY=1 -- || exit 1
run nb2_d6 TCMI_SYM_BLOCKS=2 -- --decoders 6 || exit 1
run nb2_d6_g4 TCMI_SYM_BLOCKS=2 -- --decoders 6 --gpu-streams 4 || exit 1
run nb4_d6_g4 TCMI_SYM_BLOCKS=4 -- --decoders 6 --gpu-streams 4 || exit 1
