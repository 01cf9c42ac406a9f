#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/quick_many4.py > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_run7/trace/t_kernel_trace.csv")))
rows = [r for r in rows if "rtn_group_wave" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-60:]
prev_end = None
for r in last[:40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Grid_Size_X"], r["Grid_Size_Y"], "dur us", round((e - s) / 1e3, 1), "gap us", None if prev_end is None else round((s - prev_end) / 1e3, 1))
    prev_end = e
PY
