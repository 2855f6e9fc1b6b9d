#!/bin/bash
# tools/pmc_run.sh OUTDIR [bench args...] — rocprofv3 counter passes over a short HBM-resident bench leg (one --pmc set
# per pass, kernel-trace only, as MI355X_MICROARCH.md prescribes), then tools/pmc_summary.py prints the per-kernel averages.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  d=$out/pmc_$(echo $set | cut -c1-16 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 bench.py --only-resident --resident-steps 12 "$@" > $d.json 2> $d.err || { echo "pass failed: $set"; tail -5 $d.err; exit 1; }
done
python3 tools/pmc_summary.py $out/pmc_*
