# per-queue timeline of a rocprofv3 --kernel-trace CSV: busy time, gaps between consecutive kernels of a queue, overlap across queues
#   python3 tools/trace_gaps.py DIR
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28]))
t_all0 = min(int(r["Start_Timestamp"]) for r in rows); t_all1 = max(int(r["End_Timestamp"]) for r in rows)
print("kernels", len(rows), "queues", len(byq), "span ms", (t_all1 - t_all0) / 1e6)
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:6]:
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ks[:-1], ks[1:]):
        gaps[n0 + " -> " + n1].append(s1 - e0)
    print("queue", q, "kernels", len(ks), "span ms %.2f busy ms %.2f (%.0f %%)" % (span / 1e6, busy / 1e6, 100.0 * busy / max(1, span)))
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("    %-60s n %5d  mean gap %8.1f us  total %8.2f ms" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
# time-weighted number of kernels running at once
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels in flight (time share):", {k: round(100.0 * v / tot, 1) for k, v in sorted(hist.items())})
