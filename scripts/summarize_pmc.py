"""Average the PMC counters of the largest launches of a kernel (rocprofv3 csv output): summarize_pmc.py <dir> <kernel substring>"""
import collections, csv, glob, sys
root, needle = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"]]
    if not rows:
        continue
    big = max(int(r["Grid_Size"]) for r in rows)
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in rows:
        if int(r["Grid_Size"]) == big:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-3], "grid", big, {k: round(v / n[k]) for k, v in agg.items()}, "launches", max(n.values()))
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if needle in r["Name"]:
            print("stats", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
