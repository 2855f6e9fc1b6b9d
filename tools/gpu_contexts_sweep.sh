#!/bin/bash
# the headline (compressed bytes resident in HBM -> FASTA) by the number of GPU contexts: what DESIGN §6 quotes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ctx
for g in 3 4 6 8 12 16; do
  timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --no-hard-bam --no-cli-batch --gpu-streams $g > gpurun_out/ctx/g$g.json 2> gpurun_out/ctx/g$g.err || { echo "g$g failed"; tail -3 gpurun_out/ctx/g$g.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ctx/g$g.json').read().strip().splitlines()[-1])
print('contexts', $g, round(d['value']/1e6,2), 'M positions/s', round(d['ms_per_step'],4), 'ms; file leg', round(d['file_to_fasta']['value']/1e6,2), d['fasta_bit_exact'])
"
done
