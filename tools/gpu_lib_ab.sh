#!/bin/bash
# tools/gpu_lib_ab.sh KIND LIB... — the decoder's kernel times on one file kind (tools/inflate_time.py) per build under trueconsense_amd/lib/var/lib<LIB>.so, turn about, twice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kind=$1; shift
for r in 1 2; do for v in "$@"; do
  echo "== $kind $v: $(TCMI_LIB=$PWD/trueconsense_amd/lib/var/lib$v.so timeout -k 10 200 python3 tools/inflate_time.py $kind 1000000 2>&1 | grep -E '^(inflate|crc|counts)' | tr '\n' ' ')"
done; done
