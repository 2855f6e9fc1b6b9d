# the short of a bench line: python3 tools/bench_summary.py FILE.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "value_file_to_fasta", "value_hard_bam", "value_real_bam", "value_configs2", "configs2_fasta_exact",
          "kernel_sum_single_stream_us", "fasta_bit_exact"):
    print(k, d.get(k))
print("all timed FASTAs equal the oracle chain:", d["fasta_all_timed"]["all_equal_the_oracle_chain"], "|", d["fasta_all_timed"]["chain"])
print("single", round(d["e2e_single_bam"]["seconds"] * 1e3, 3), "ms", {k: round(v * 1e3, 3) for k, v in d["e2e_single_bam"]["stage_seconds"].items()})
print("alone    ", {k: round(v["us_per_bam"], 1) for k, v in d["cold_kernels"].items()})
print("pipelined", {k: round(v["us_per_bam"], 1) for k, v in d["cold_kernels_pipelined"].items()})
for b in ("roofline", "roofline_hot_path"):
    r = d[b]
    print(b, "frac", round(r["frac"], 4), "single-stream", round(r["single_stream_frac"] or 0, 4), "aggregate", round(r.get("aggregate", {}).get("frac", 0), 4), "|", r["kernel"][:60])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu", round(c["value"]), c["unit"], "cores", c["cores"], "| python stage B", c.get("python_stage_b", {}).get("seconds_per_bam"), "s, stand-in",
          c.get("python_reference_standin_seconds_per_bam"), "s per BAM")
for leg in ("hard_bam", "real_bam"):
    if leg in d:
        print(leg, round(d[leg]["pipelined"]["ms_per_bam"], 3), "ms overlapped,", round(d[leg]["seconds_per_bam"] * 1e3, 2), "ms alone,", d[leg]["fasta_bit_exact"], d[leg]["kernels_us"])
if "cli_batch" in d:
    print("cli_batch ms/sample", d["cli_batch"].get("ms_per_bam"), d["cli_batch"].get("error"))
if "resident" in d:
    print("resident", round(d["resident"]["value"] / 1e6), "M/s, tally frac", round(d["resident"]["roofline"]["frac"], 3))
if "configs2" in d:
    print("configs2", round(d["configs2"]["value"] / 1e6, 2), "M/s", d["configs2"]["ms_per_bam"], d["configs2"]["fasta_all_timed"]["all_equal_the_oracle_chain"], d["configs2"]["consensus_len"])
