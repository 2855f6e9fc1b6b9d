#!/bin/bash
# phase clocks of bgzf_symbols / bgzf_copy on the bench file, the harder one and the real-data-like one, from a build with
# -DTCMI_COPY_PHASES=1 (trueconsense_amd/lib/var/libph1.so) and =2 (libph2.so); usage: tools/gpu_phases.sh [kinds...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ph
kinds=${@:-headline hard real}
for kind in $kinds; do for v in 1 2; do
  lib=$PWD/trueconsense_amd/lib/var/libph$v.so
  [ -f $lib ] || continue
  echo "== $kind, TCMI_COPY_PHASES=$v"
  TCMI_COPY_PHASES=$v TCMI_LIB=$lib timeout -k 10 200 python3 tools/inflate_stamps.py $kind 1000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ph/${kind}_$v.log || exit 1
done; done
