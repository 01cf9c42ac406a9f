#!/usr/bin/env python3
"""Turn a scripts/profile_r03.sh output directory into the small files copied to profiles/r03_*."""
import collections, csv, glob, json, os, statistics, sys

root = sys.argv[1]
out = os.path.join(root, "summary")
os.makedirs(out, exist_ok=True)


def find(d, pat):
    return sorted(glob.glob(os.path.join(root, d, "**", pat), recursive=True))


def short(name):
    return name.split("(")[0].replace("void ", "")


def kernel_table(d, skip_first=0, only=None, top=14):
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    table = []
    for k, v in acc.items():
        if only and not any(o in k for o in only):
            continue
        v2 = v[skip_first:] if len(v) > 2 * skip_first + 4 else v
        table.append({"kernel": short(k) if len(k) > 160 else k, "calls": len(v), "total_ms": round(sum(v) / 1e6, 3),
                      "avg_us": round(statistics.mean(v2) / 1e3, 3), "median_us": round(statistics.median(v2) / 1e3, 3),
                      "min_us": round(min(v2) / 1e3, 3), "p90_us": round(sorted(v2)[len(v2) * 9 // 10] / 1e3, 3)})
    table.sort(key=lambda t: -t["total_ms"])
    span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e6 if rows else 0
    return {"span_ms": round(span, 2), "kernels": table[:top]}


def counters(d, kern):
    res = {}
    for f in find(d, "*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = {"avg_per_launch": statistics.mean(v[10:] if len(v) > 30 else v), "launches": len(v)}
    return res


def copy_stats(d, dst):
    fs = find(d, "*kernel_stats.csv")
    if fs:
        lines = open(fs[0]).read().splitlines()
        keep = [lines[0]] + [l for l in lines[1:] if "oq::" in l][:12]
        open(os.path.join(out, dst), "w").write("\n".join(l[:400] for l in keep) + "\n")


# ---- 1. RTN
alg = 4096 * 11008 * 4 + 4096 * 11008 // 2 + 352256 * 5
traffic = {}
for lay, kern in (("nbits", "rtn_group_wave"), ("kn", "rtn_group_fused")):
    t = kernel_table(f"rtn_{lay}/trace", skip_first=10, only=["rtn_group", "transpose_qparams"])
    c = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        c.update(counters(f"rtn_{lay}/{sub}", kern))
    rec = {"command": f"rocprofv3 ... -- python3 bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --layout {lay}",
           "kernels": t["kernels"], "counters": c, "algorithmic_bytes_per_launch": alg}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE in KiB counts 1/2 of a wide coalesced stream on gfx950; WRITE_SIZE in KiB is exact
        rd = c["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
        wr = c["WRITE_SIZE"]["avg_per_launch"] * 1024
        rec["hbm_read_bytes_corrected"] = rd
        rec["hbm_write_bytes"] = wr
        traffic[lay] = int(rd + wr)
    main = next((k for k in t["kernels"] if kern in k["kernel"]), None)
    if main:
        key = "" if lay == "nbits" else "_main_kernel_only"          # the [K,N] route has a second launch (transpose_qparams)
        rec["achieved_GBs_algorithmic_avg" + key] = round(alg / (main["avg_us"] * 1e-6) / 1e9, 1)
        rec["frac_of_8TBs" + key] = round(alg / (main["avg_us"] * 1e-6) / 1e9 / 8000, 4)
        tr = next((k for k in t["kernels"] if "transpose_qparams" in k["kernel"]), None)
        if lay == "kn" and tr:
            rec["frac_of_8TBs_both_kernels_by_duration_sum"] = round(alg / ((main["avg_us"] + tr["avg_us"]) * 1e-6) / 1e9 / 8000, 4)
    json.dump(rec, open(os.path.join(out, f"r03_rtn_{lay}_rocprof.json"), "w"), indent=1)
    copy_stats(f"rtn_{lay}/trace", f"r03_rtn_{lay}_kernel_stats.csv")
if traffic:
    json.dump({**traffic, "source": "profiles/r03_rtn_{nbits,kn}_rocprof.json (round 3: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
               "`bench.py --no-extras`, FETCH_SIZE x2 gfx950 correction, per launch of the dominant kernel; kn excludes the 1.7 MB transpose_qparams launch)",
               "note": "HBM bytes per launch from rocprofv3 PMC passes: FETCH_SIZE*1024*2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024"},
              open(os.path.join(out, "rtn_pmc_traffic.json"), "w"), indent=1)

# ---- 2. GPTQ (parity + corrected pass of 8 layers)
g = kernel_table("gptq/trace", top=24)
g["command"] = "rocprofv3 --kernel-trace --stats -- python3 bench_gptq.py --layers 8 --no-cpu-baseline --hessian-methods '' --extra-passes corrected"
g["note"] = ("8 of 32 Llama-2-7B layers; one process = warm-up + the parity pass + the corrected pass + the verification launches after the timed regions; "
             "kernels serialised by the profiler (no two-stream overlap)")
json.dump(g, open(os.path.join(out, "r03_gptq_kernels.json"), "w"), indent=1)
copy_stats("gptq/trace", "r03_gptq_kernel_stats.csv")

# ---- 3. corrected loop / Hessian
l = kernel_table("loop/trace", top=16, only=["gptq_rows16", "panel_update", "gemm_f16x3", "gemm_tn", "split_f16x2", "absmax", "gptq_coef_image"])
l["command"] = "rocprofv3 --kernel-trace --stats -- python3 scripts/quick_loop.py   (corrected loop alone: 4096x4096, 4096x11008, 11008x4096; 4 calls each)"
l["counters_gptq_rows16_kernel"] = {**counters("loop/pmc_sq", "gptq_rows16"), **counters("loop/pmc_mfma", "gptq_rows16")}
l["counters_gemm_f16x3_kernel"] = counters("loop/pmc_mfma", "gemm_f16x3")
json.dump(l, open(os.path.join(out, "r03_gptq_loop_kernels.json"), "w"), indent=1)
h = kernel_table("hess/trace", top=8, only=["syrk", "split", "absmax"])
h["command"] = "rocprofv3 ... -- python3 scripts/quick_hessian.py f16x3 11008   (K = 11008, 65 536 rows per call)"
h["counters_syrk_f16_m16_kernel"] = counters("hess/pmc_mfma", "syrk_f16_m16")
c = h["counters_syrk_f16_m16_kernel"]
k = next((x for x in h["kernels"] if "syrk_f16_m16" in x["kernel"]), None)
if k and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 1024 SIMDs
    cyc = c["GRBM_GUI_ACTIVE"]["avg_per_launch"] / 8
    h["derived"] = {"clock_GHz": round(cyc / (k["avg_us"] * 1e3), 3), "mfma_pipe_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_per_launch"] / 1024 / cyc, 4)}
json.dump(h, open(os.path.join(out, "r03_hessian_f16x3_kernels.json"), "w"), indent=1)

f = kernel_table("factor/trace", top=16, only=["chol_diag", "gemm_tn", "gemm_f16x3", "syrk_f16_m16_many", "split_f16x2_many", "absmax", "reverse_copy",
                                                "finish_factor", "place_diag", "plan"])
f["command"] = ("rocprofv3 --kernel-trace --stats -- python3 scripts/lab_factor_trace.py 11008 8   (two batched factor chains of 8 Hessians of 11008: "
                "halve the totals for one chain)")
json.dump(f, open(os.path.join(out, "r03_factor_kernels.json"), "w"), indent=1)
hm = kernel_table("hess_many/trace", top=10, only=["_many_kernel", "syrk", "split", "absmax", "gemm_tn", "plan"])
hm["command"] = ("rocprofv3 --kernel-trace --stats -- python3 scripts/quick_hessian_many.py   (72 gemma-3-270m-shaped inputs per batch: 12 batches per route, "
                 "per-tensor calls on one and on four streams, then the grouped call)")
json.dump(hm, open(os.path.join(out, "r03_hessian_many_kernels.json"), "w"), indent=1)

# ---- 4. calibration / AWQ
cal = kernel_table("calib/trace", top=8, only=["minmax", "rtn_many", "qparams"])
cal["command"] = "rocprofv3 --kernel-trace --stats -- python3 bench_calib.py --no-cpu-baseline"
json.dump(cal, open(os.path.join(out, "r03_calibration_kernels.json"), "w"), indent=1)
aw = kernel_table("awq/trace", top=14)
aw["command"] = "rocprofv3 --kernel-trace --stats -- python3 scripts/quick_awq.py   (4096^3 layer: 4 scale searches, 4 clip searches, 4 smoothing scales)"
json.dump(aw, open(os.path.join(out, "r03_awq_kernels.json"), "w"), indent=1)
print("summaries:", sorted(os.listdir(out)))
