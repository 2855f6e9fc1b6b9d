import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "tally" not in k and "call" not in k: continue
            print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
