cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/pmc3; mkdir -p $d
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
n=$(echo $set | cut -c1-16 | tr ' ' '_')
rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d/$n -- python3 bench.py --no-resident --no-cpu-baseline --steps 8 --warmup 2 --min-seconds 0 --gpu-streams 1 > $d/$n.json 2> $d/$n.err || { tail -5 $d/$n.err; exit 1; }
done
python3 - $d/* <<'PY'
import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][22:52].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "inflate" in k: print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in cs.items()})
PY
