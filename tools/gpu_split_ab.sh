#!/bin/bash
# tools/gpu_split_ab.sh REPS READS VALUE... — bench.py --split-bam --from-file (one file, world 1, the RCCL hook) per value of TCMI_SPLIT_SUB, turn about on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
reps=$1; reads=$2; shift 2
for r in $(seq $reps); do for v in "$@"; do
  TCMI_SPLIT_SUB=$v timeout -k 10 300 python3 bench.py --split-bam --from-file --reads $reads --steps 20 --warmup 3 > gpurun_out/ab/split_$v.json 2> gpurun_out/ab/split_$v.err || { echo "$v: fail"; tail -3 gpurun_out/ab/split_$v.err; exit 1; }
  echo "TCMI_SPLIT_SUB=$v reads=$reads: $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab/split_$v.json').read().strip().splitlines()[-1]); print('ms/step %.3f in tcmi_split_step %.3f median %.3f fasta %s counts %s' % (d['ms_per_step'], d['ms_per_step_in_tcmi_split_step'], d['ms_per_step_median'], d['fasta_bit_exact'], d['counts_bit_exact']))")"
done; done
