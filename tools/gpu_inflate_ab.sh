#!/bin/bash
# tools/gpu_inflate_ab.sh [kinds...] — the decoder's GPU tests on the default build, then kernel times of the decoder per file kind
# with trueconsense_amd/lib/var/libbase.so (a build of the commit before) and the default build, turn about on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/iab
timeout -k 10 500 python -m pytest tests/test_bam_device.py tests/test_bam_fixture.py tests/test_one_sync.py -x -q -m gpu > gpurun_out/iab/tests.log 2>&1 || { tail -40 gpurun_out/iab/tests.log; exit 1; }
tail -2 gpurun_out/iab/tests.log
kinds=${@:-headline hard real}
for kind in $kinds; do for r in 1 2; do for v in base new; do
  lib=$PWD/trueconsense_amd/lib/libtcmi.so; [ $v = base ] && lib=$PWD/trueconsense_amd/lib/var/libbase.so
  echo "== $kind $v: $(TCMI_LIB=$lib timeout -k 10 200 python3 tools/inflate_time.py $kind 1000000 2>&1 | grep -E '^(inflate|crc|counts)' | tr '\n' ' ')"
done; done; done
