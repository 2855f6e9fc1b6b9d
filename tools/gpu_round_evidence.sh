#!/bin/bash
# the > 4 GiB single-file run of a round (tools/big_bam.py 16), log under gpurun_out/TAG  (tools/profile_round.sh TAG is the other call)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 1000 python3 tools/big_bam.py 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/big_bam.log
