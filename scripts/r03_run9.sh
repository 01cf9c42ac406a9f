#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/quick_loop.py > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, os, collections
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_run9/trace/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last loop call of the first shape: find coef image kernels as delimiters
idx = [i for i, r in enumerate(rows) if "gptq_coef_image" in r["Kernel_Name"]]
for call, (a, b) in enumerate(zip(idx, idx[1:] + [len(rows)])):
    if call not in (3, 7, 11): continue
    seg = rows[a:b]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        acc[n][0] += 1; acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in acc.values())
    print(f"call {call}: span {span:.0f} us, busy {busy:.0f} us")
    for n, v in sorted(acc.items(), key=lambda kv: -kv[1][1]): print(f"   {n:42s} x{v[0]:4d} {v[1]:8.1f} us")
PY
