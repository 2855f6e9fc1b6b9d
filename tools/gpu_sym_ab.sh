#!/bin/bash
# blocks per workgroup of bgzf_symbols (TCMI_SYM_BLOCKS = 4 / 2 / 1): kernel times and phase clocks on both files
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/inf
for nbw in 4 2 1; do for kind in headline hard; do
  echo "== TCMI_SYM_BLOCKS=$nbw $kind"
  TCMI_SYM_BLOCKS=$nbw timeout -k 10 300 python3 tools/inflate_stamps.py $kind 1000000 2>&1 | grep -v "amdgpu.ids\|^crc\|^records\|file MB" | tee gpurun_out/inf/ab_${nbw}_$kind.log || exit 1
done; done
