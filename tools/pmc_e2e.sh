#!/bin/bash
# tools/pmc_e2e.sh OUTDIR [bench args...] — rocprofv3 counter passes (one --pmc set per pass, kernel-trace only) over a short
# file -> FASTA bench leg; prints per-kernel averages for the cold-path kernels.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  d=$out/pmc_$(echo $set | cut -c1-16 | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 bench.py --no-resident --no-cpu-baseline --no-hard-bam --no-cli-batch --no-configs0 --steps 8 --warmup 2 --min-seconds 0 --gpu-streams 1 "$@" > $d.json 2> $d.err || { echo "pass failed: $set"; tail -5 $d.err; exit 1; }
done
python3 - $out/pmc_* <<'PY'
import csv, sys, glob, collections, re
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            m = re.search(r"::(\w+)(<[^>]*>)?\(", row["Kernel_Name"])
            k = (m.group(1) + (m.group(2) or "")) if m else row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if not any(x in k for x in ("bgzf_", "pk_", "tally", "call_kernel", "rec_", "ins_")): continue
            print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
