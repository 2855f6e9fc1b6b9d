#!/bin/bash
# tools/profile_round.sh TAG — the evidence set of a round, written under gpurun_out/TAG (copy what is to be judged into
# profiles/): the default bench plain, the HBM-resident leg under rocprofv3 --kernel-trace --stats, PMC passes, the
# configs[2] / configs[4] shapes.
tag=$1
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
set -e
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_resident -- python3 bench.py --only-resident > $out/bench_resident_under_rocprof.json 2> $out/rocprof_resident.err
bash tools/pmc_run.sh $out/pmc > $out/pmc_summary.txt 2>&1
python3 bench.py --no-cpu-baseline --indels --steps 8 --warmup 2 > $out/bench_indels.json 2> $out/bench_indels.err
python3 bench.py --split-bam --steps 100 --warmup 10 > $out/bench_split.json 2> $out/bench_split.err
echo done
