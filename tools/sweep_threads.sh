#!/bin/bash
# tools/sweep_threads.sh "D S" ["D S" ...] — file -> FASTA throughput for reader-thread / GPU-context counts, on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sw
for cfg in "$@"; do set -- $cfg
  timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --decoders $1 --gpu-streams $2 > gpurun_out/sw/d$1s$2.json 2> gpurun_out/sw/d$1s$2.err || { echo fail $cfg; tail -3 gpurun_out/sw/d$1s$2.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw/d$1s$2.json').read().strip().splitlines()[-1])
print('decoders $1 streams $2:', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,2) for k,v in d['e2e_stage_busy_seconds_per_bam'].items()})
"
done
