#!/bin/bash
# tools/profile_round.sh TAG — the evidence set of a round, written under gpurun_out/TAG (copy what is to be judged into profiles/):
# default bench plain and under rocprofv3 --kernel-trace --stats, one-BAM-per-launch under rocprofv3, PMC passes, the
# configs[2] / configs[4] shapes, the command line end to end.
tag=$1
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
set -e
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_batch1 -- python3 bench.py --no-cpu-baseline --batch 1 --bams 4 --steps 1000 > $out/bench_batch1_under_rocprof.json 2> $out/rocprof_batch1.err
bash tools/pmc_run.sh $out/pmc > $out/pmc_summary.txt 2>&1
python3 bench.py --no-cpu-baseline --indels --steps 200 --warmup 20 > $out/bench_indels.json 2> $out/bench_indels.err
python3 bench.py --no-cpu-baseline --split-bam --steps 100 --warmup 10 > $out/bench_split.json 2> $out/bench_split.err
python3 tools/e2e_cli.py 1000000 > $out/e2e_cli.json 2> $out/e2e_cli.err
echo done
