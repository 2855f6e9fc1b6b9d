#!/bin/bash
# the decoder's GPU tests, then kernel times (and phase clocks) of the decoder on the bench file and on the harder one
# usage: tools/gpu_inflate_check.sh [legacy]   (legacy: also time round 2's one-kernel decoder)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/inf
timeout -k 10 400 python -m pytest tests/test_bam_device.py tests/test_bam_fixture.py -x -q -m gpu > gpurun_out/inf/tests.log 2>&1 || { tail -40 gpurun_out/inf/tests.log; exit 1; }
tail -3 gpurun_out/inf/tests.log
for kind in headline hard; do
  timeout -k 10 300 python3 tools/inflate_stamps.py $kind 1000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/inf/new_$kind.log || exit 1
  if false; then
    TCMI_INFLATE_LEGACY=1 timeout -k 10 300 python3 tools/inflate_time.py $kind 1000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/inf/legacy_$kind.log || exit 1
  fi
done
