#!/bin/bash
# kernel trace of the model-scale RTN call: where the time between the first and the last kernel of a call goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/model_rtn_trace; rm -rf $OUT; mkdir -p $OUT
python3 $R/scripts/lab_model_rtn.py > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/lab_model_rtn.py > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/model_rtn_trace"
print(open(root + "/plain.txt").read())
f = glob.glob(f"{root}/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"]) for r in csv.DictReader(open(f))))
rtn = [r for r in rows if "rtn_group_wave" in r[2]]
# split into calls: a gap of more than 2 ms between rtn kernels starts a new call
calls, cur = [], [rtn[0]]
for a, b in zip(rtn, rtn[1:]):
    if b[0] - a[1] > 2_000_000: calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
for c in calls[-2:]:
    span = (c[-1][1] - c[0][0]) / 1e3
    busy = sum(e - s for s, e, *_ in c) / 1e3
    gaps = [((b[0] - a[1]) / 1e3, i) for i, (a, b) in enumerate(zip(c, c[1:]))]
    big = [(round(g, 1), i, c[i][3], c[i][4], c[i + 1][3], c[i + 1][4]) for g, i in gaps if g > 3.0]
    print("launches", len(c), "span_us", round(span), "kernel_sum_us", round(busy), "gaps_us", round(span - busy), "gaps > 3 us:", big)
    byshape = {}
    for s, e, nme, gx, gy in c:
        byshape.setdefault((gx, gy), []).append((e - s) / 1e3)
    print({k: (len(v), round(sum(v) / len(v), 1)) for k, v in byshape.items()})
PY
