#!/bin/bash
# tools/gpu_hard_ab.sh REPS LIB... — the bench's headline + hard_bam + real_bam legs per build, turn about on one box (see tools/gpu_ab.sh for LIB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
lib_of() { [ "$1" = default ] && echo $PWD/trueconsense_amd/lib/libtcmi.so || echo $PWD/trueconsense_amd/lib/var/lib$1.so; }
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  TCMI_LIB=$(lib_of $v) timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-batch --no-configs2 --no-configs0 --no-resident $BENCH_ARGS \
      > gpurun_out/ab/hard_$v.json 2> gpurun_out/ab/hard_$v.err || { echo "$v: fail"; tail -3 gpurun_out/ab/hard_$v.err; exit 1; }
  echo "$v: $(python3 tools/bench_summary.py gpurun_out/ab/hard_$v.json 2>/dev/null | grep -E '^(value|value_hard_bam|value_real_bam|hard_bam|real_bam|fasta_bit_exact) ' | tr '\n' ' ')"
done; done
