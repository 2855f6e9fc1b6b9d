#!/bin/bash
# tools/ab_env.sh VAR VALUE_A VALUE_B [reps] — file -> FASTA throughput with an environment variable at two values, alternating on one box
var=$1; va=$2; vb=$3; reps=${4:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abe
for r in $(seq $reps); do for v in "$va" "$vb"; do
  env $var=$v timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline > gpurun_out/abe/$v.json 2> gpurun_out/abe/$v.err || { echo fail; tail -5 gpurun_out/abe/$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/abe/$v.json').read().strip().splitlines()[-1])
print('$var=$v', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), {k:round(x*1e3,2) for k,x in d['e2e_stage_busy_seconds_per_bam'].items()}, 'single', round(d['e2e_single_bam']['seconds']*1e3,2), {k:round(x*1e3,2) for k,x in d['e2e_single_bam']['stage_seconds'].items()}, d['fasta_bit_exact'])
"
done; done
