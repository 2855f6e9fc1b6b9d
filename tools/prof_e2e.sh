#!/bin/bash
# tools/prof_e2e.sh OUTDIR [bench args...] — rocprofv3 --kernel-trace --stats over a short file -> FASTA bench leg (every kernel
# (three GPU contexts: with five or more — the bench runs eight by default — rocprofv3 7.2 segfaults in its own copy of an API record, twice out of twice)
# of the cold path: bgzf_symbols, bgzf_copy, pk_*, tally_planes_kernel, call_kernel — of the bench's own 1M-read files only: the configs[2] / configs[0]
# legs are left out, their files' kernels would be averaged in); the stats CSV lands in OUTDIR.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-resident --no-cpu-baseline --no-cli-batch --no-hard-bam --no-configs2 --no-configs0 --gpu-streams 3 --steps 16 --warmup 2 "$@" > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
f=$(ls $out/stats/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
cat $out/kernel_stats.csv
