cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tb in 0 1536 4096 100000; do
  TCMI_TEAM_BYTES=$tb timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --min-seconds 0 --no-resident --no-cpu-baseline --no-cli-batch > gpurun_out/hb_$tb.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/hb_$tb.json').read().strip().splitlines()[-1])
print($tb, d['hard_bam']['kernels_us'], d['hard_bam']['fasta_bit_exact'])
"
done
