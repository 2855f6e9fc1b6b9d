#!/bin/bash
# tools/gpu_window_ab2.sh kind [window bytes...] — bgzf_symbols' window (TCMI_SYM_WINDOW; 0 = payload staged whole) on one file kind
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kind=${1:-real}; shift
for w in ${@:-6144 5120 4096 3072}; do
  echo "== $kind window=$w: $(TCMI_SYM_WINDOW=$w timeout -k 10 300 python3 tools/inflate_time.py $kind 1000000 2>&1 | grep -E '^(inflate|counts)' | tr '\n' ' ')"
done
