#!/bin/bash
# A/B of library variants on one box, default batched bench (HIP-event kernel time + whole-job rate)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  TCMI_LIB=$GRAFT_REPO_ROOT/trueconsense_amd/lib/variants/libtcmi_$v.so python3 bench.py --no-cpu-baseline --steps 400 > /tmp/ab_$v.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/ab_$v.json')); print('$v rep$rep tally_us=%.1f value=%.0fM frac=%.3f' % (d['kernels_us']['tally'], d['value']/1e6, d['roofline']['frac']))"
done
done
