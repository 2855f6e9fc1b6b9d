cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
for cfg in "4 3" "8 3" "8 4" "8 5" "4 3" "8 4"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python3 bench.py --no-resident --no-cpu-baseline --gpu-streams $2 > gpurun_out/q/q$1s$2.json 2> gpurun_out/q/q$1s$2.err || { echo fail $cfg; tail -3 gpurun_out/q/q$1s$2.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/q/q$1s$2.json').read().strip().splitlines()[-1])
print('hwq $1 streams $2:', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,2) for k,v in d['e2e_stage_busy_seconds_per_bam'].items()})
"
done
