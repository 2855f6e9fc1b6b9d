"""tools/pmc_e2e.sh's summary (per-kernel PMC averages, one 1M-read BAM per launch) -> the "round6" block (or the key given as third argument: round6_hard, round6_real) of profiles/traffic.json that
bench.py's roofline blocks quote.   python3 tools/pmc_to_traffic.py gpurun_out/TAG/pmc_e2e_summary.txt profiles/TAG_pmc_e2e.txt"""
import ast, json, os, re, sys

src, committed_as = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "round6"
k, seen = {}, {}
for line in open(src):
    m = re.match(r"^([\w<>, ]+?) (\{.*\}) n= (\d+)$", line.strip())
    if not m:
        continue
    name = re.sub(r"<.*>", "", m.group(1))
    vals, n = ast.literal_eval(m.group(2)), int(m.group(3))
    # (a pass writes a file per process: the bench's own — the most launches — counts, not the command line's child processes)
    for c, v in vals.items():
        if n >= seen.get((name, c), 0):
            seen[(name, c)] = n
            k.setdefault(name, {})[c] = v
# FETCH_SIZE: KiB as reported; gfx950 tallies the 128-byte requests of a wide coalesced stream (16 bytes per lane, consecutive lanes) at
# 64 bytes (MI355X_MICROARCH.md): doubled for the kernels that read that way
wide = {"tally_planes_kernel"}
out = {}
for name, c in k.items():
    if "FETCH_SIZE" not in c:
        continue
    insts = {x: round(c.get("SQ_INSTS_" + x.upper(), 0)) for x in ("valu", "salu", "lds", "smem")}
    insts["vmem"] = round(c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0))
    out[name] = {"fetch_kib": c["FETCH_SIZE"], "fetch_doubled": name in wide, "write_kib": c.get("WRITE_SIZE", 0.0),
                 "hbm_bytes": round((c["FETCH_SIZE"] * (2 if name in wide else 1) + c.get("WRITE_SIZE", 0.0)) * 1024),
                 "wave_insts": insts, "wave_insts_total": sum(insts.values()), "waves": round(c.get("SQ_WAVES", 0)),
                 "lds_bank_conflict_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0))}
    # where the wave-cycles go: a wave that neither issues (ACTIVE_INST_ANY) nor waits in the issue stage for its instruction's unit
    # (WAIT_INST_ANY) is parked at an s_waitcnt or a barrier — on memory, mostly
    wc, act, wait = c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_ACTIVE_INST_ANY", 0.0), c.get("SQ_WAIT_INST_ANY", 0.0)
    if wc > 0:
        out[name].update(wave_cycles=round(wc), active_inst_any=round(act), wait_inst_any=round(wait), parked=round(1.0 - (act + wait) / wc, 4))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tp = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(tp))
t[key] = {"source": committed_as + " (tools/pmc_e2e.sh: one --pmc set per pass, kernel trace only; one 1M-read BAM per launch, single stream)",
               "kernels": out, "wave_insts_per_bam": sum(v["wave_insts_total"] for v in out.values()),
               "hbm_bytes_per_bam": sum(v["hbm_bytes"] for v in out.values())}
json.dump(t, open(tp, "w"), indent=1)
if "tally_planes_kernel" in out:
    t[key]["tally_hbm_bytes_per_bam"] = out["tally_planes_kernel"]["hbm_bytes"]
print(json.dumps(t[key], indent=1)[:1500])
