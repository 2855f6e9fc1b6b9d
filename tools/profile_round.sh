#!/bin/bash
# tools/profile_round.sh TAG — the evidence set of a round, written under gpurun_out/TAG (copy what is to be judged into
# profiles/): the default bench, the driver's shape (--steps 20 --warmup 5), the file -> FASTA leg under rocprofv3
# --kernel-trace --stats, its PMC passes, the HBM-resident leg under rocprofv3 (tally kernel), the configs[2] / [4] shapes, the
# kernels alone on the GPU.  (tools/gpu_round_evidence.sh TAG: the > 4 GiB single-file run, a call of its own.)
tag=$1
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
set -e
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "default done"
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
echo "driver shape done"
bash tools/prof_e2e.sh $out/e2e > $out/e2e_kernel_stats.txt 2>&1
echo "e2e stats done"
bash tools/pmc_e2e.sh $out/pmc_e2e > $out/pmc_e2e_summary.txt 2>&1
echo "e2e pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_resident -- python3 bench.py --only-resident > $out/bench_resident_under_rocprof.json 2> $out/rocprof_resident.err
cp $(ls $out/stats_resident/*/*kernel_stats.csv | head -1) $out/resident_kernel_stats.csv
echo "resident stats done"
python3 bench.py --no-cpu-baseline --no-resident --no-cli-batch --indels --steps 8 --warmup 2 > $out/bench_indels.json 2> $out/bench_indels.err
echo "indels done"
python3 bench.py --split-bam --steps 100 --warmup 10 > $out/bench_split.json 2> $out/bench_split.err
python3 bench.py --split-bam --from-file --reads 4000000 --steps 20 --warmup 3 > $out/bench_split_from_file.json 2> $out/bench_split_from_file.err
echo "split done"
python3 bench.py --host-decode --no-cpu-baseline --no-resident --steps 16 > $out/bench_host_decode.json 2> $out/bench_host_decode.err
echo "host decode done"
# the kernels with nothing else on the GPU (one context, one reader)
bash tools/prof_e2e.sh $out/single --gpu-streams 1 --decoders 1 --steps 8 --min-seconds 0.1 > $out/single.log 2>&1
echo done
