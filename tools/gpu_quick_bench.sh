#!/bin/bash
# the headline legs only (no CPU baseline, no resident / hard / command-line legs): value, file leg, kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=${1:-q}; shift
timeout -k 10 300 python3 bench.py --no-resident --no-cpu-baseline --no-hard-bam --no-cli-batch "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err || { tail -5 gpurun_out/$name.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms', d['fasta_bit_exact'], d['fasta_all_timed']['all_equal_the_oracle_chain'], 'file leg', round(d['file_to_fasta']['value']/1e6,2))
print('  alone    ', {k:round(v['us_per_bam'],1) for k,v in d['cold_kernels'].items()})
print('  pipelined', {k:round(v['us_per_bam'],1) for k,v in d['cold_kernels_pipelined'].items()})
"
