#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory (trace/ + pmc_*/ sub-directories of separate runs) into one JSON: per kernel
calls / average / median / min / p90 duration, and per kernel the average counter values per launch, with the gfx950
corrections of MI355X_MICROARCH.md (HBM): FETCH_SIZE (KiB) counts half of a wide coalesced read stream, WRITE_SIZE (KiB)
is exact.  usage: summarize_kernels.py <dir> [substring filter ...]"""
import collections, csv, glob, json, os, statistics, sys

root = sys.argv[1]
only = sys.argv[2:]


def short(name):
    return name.split("(")[0].replace("void ", "")[:120]


rows = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
kern = {}
for k, v in acc.items():
    if only and not any(o in k for o in only):
        continue
    v2 = v[5:] if len(v) > 20 else v
    kern[k] = {"calls": len(v), "total_ms": round(sum(v) / 1e6, 3), "avg_us": round(statistics.mean(v2) / 1e3, 3),
               "median_us": round(statistics.median(v2) / 1e3, 3), "min_us": round(min(v2) / 1e3, 3),
               "p90_us": round(sorted(v2)[len(v2) * 9 // 10] / 1e3, 3)}
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in ctr.items():
    if k not in kern:
        continue
    c = {n: statistics.mean(v[5:] if len(v) > 20 else v) for n, v in cs.items()}
    kern[k]["counters_avg_per_launch"] = c
    if "FETCH_SIZE" in c:
        kern[k]["hbm_read_MB_corrected"] = round(c["FETCH_SIZE"] * 1024 * 2 / 1e6, 2)
    if "WRITE_SIZE" in c:
        kern[k]["hbm_write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 2)
print(json.dumps({"dir": os.path.basename(root.rstrip("/")), "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"]))}, indent=1))
