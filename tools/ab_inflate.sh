#!/bin/bash
# tools/ab_inflate.sh [variant letters...] — A/B of bgzf_inflate builds (trueconsense_amd/lib/var/lib<V>.so, built by hand with
# different -DTCMI_INFLATE_* values): the decoder's GPU tests on the default build, then the file -> FASTA bench per variant.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests/test_bam_device.py tests/test_bam_fixture.py -x -q -m gpu > gpurun_out/ab/tests.log 2>&1 || { tail -20 gpurun_out/ab/tests.log; exit 1; }
tail -2 gpurun_out/ab/tests.log
for v in "$@"; do
  TCMI_INFLATE_OCCUPANCY=1 TCMI_LIB=$PWD/trueconsense_amd/lib/var/lib$v.so timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --steps 24 --warmup 4 > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "variant $v failed"; tail -5 gpurun_out/ab/$v.err; exit 1; }
  grep -h "wavefronts per CU" gpurun_out/ab/$v.err | head -1
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab/$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), 'M/s', d['ms_per_step'], 'cold', {k:round(v['us_per_bam']) for k,v in d['cold_kernels'].items()}, 'pipe', {k:round(v['us_per_bam']) for k,v in d['cold_kernels_pipelined'].items()}, d['fasta_bit_exact'])
"
done
for gs in $STREAMS; do
  timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --gpu-streams $gs > gpurun_out/ab/gs$gs.json 2> gpurun_out/ab/gs$gs.err || { echo "streams $gs failed"; tail -5 gpurun_out/ab/gs$gs.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab/gs$gs.json').read().strip().splitlines()[-1])
print('streams $gs', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), d['e2e_stage_busy_seconds_per_bam'])
"
done
