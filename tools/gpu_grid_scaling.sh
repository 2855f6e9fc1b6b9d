#!/bin/bash
# one BAM of m million reads decoded + packed + tallied in ONE call, single stream: what a larger grid does to the time per million reads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${@:-1 2 4 8}; do
  echo "== m=$m: $(timeout -k 10 300 python3 tools/big_bam.py $m 2>&1 | grep -E '^second call|^counts equal' | tr '\n' ' ')"
done
