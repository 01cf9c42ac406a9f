#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/factor_trace; mkdir -p $OUT
for cfg in "11008 8" "4096 24"; do
  set -- $cfg
  rocprofv3 --kernel-trace --output-format csv -d $OUT/k$1 -- python3 $R/scripts/lab_factor_trace.py $1 $2 > $OUT/k$1.log 2>&1
  tail -1 $OUT/k$1.log
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/factor_trace"
for k in ("k11008", "k4096"):
    f = glob.glob(f"{root}/{k}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    # the last chain = second half of the factor kernels
    fac = [r for r in rows if any(t in r["Kernel_Name"] for t in ("gemm_tn_kernel", "chol_diag", "place_diag", "finish_factor", "reverse_copy", "damp", "_many_kernel", "factor_plan", "inverse_level_plan"))]
    fac = fac[len(fac) // 2:]
    agg = collections.OrderedDict()
    for r in fac:
        name = r["Kernel_Name"].split("(")[0][:40]
        key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
    tot = sum(a[1] for a in agg.values())
    span = (int(fac[-1]["End_Timestamp"]) - int(fac[0]["Start_Timestamp"])) / 1e3
    print(k, "kernels", len(fac), "sum_us", round(tot), "span_us", round(span))
    big = sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]
    for key, (n, t) in big:
        print("   ", key, n, round(t), "us")
    by = collections.Counter()
    for (name, *_), (n, t) in agg.items(): by[name] += t
    print("   by kernel:", {k2: round(v) for k2, v in by.items()})
PY
