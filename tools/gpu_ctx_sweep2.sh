#!/bin/bash
# headline at several context counts: tools/gpu_ctx_sweep2.sh TAG "3 4 6 8 12" [env...]
tag=$1; shift
list=$1; shift
for n in $list; do
  bash tools/gpu_quick_bench.sh ${tag}_c$n --steps 20 --warmup 5 --gpu-streams $n "$@" | head -1
done
