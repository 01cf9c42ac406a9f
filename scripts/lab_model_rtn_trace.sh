#!/bin/bash
# kernel trace of the model-scale RTN call (bench.py model_rtn): launch durations by grid, against the bytes each moves
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/model_rtn_trace; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gptq --no-seam --no-awq --no-calibration > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/model_rtn_trace"
f = glob.glob(f"{root}/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rtn_group_wave" in r["Kernel_Name"]]
agg = collections.OrderedDict()
for r in rows:
    key = (r["Grid_Size_X"], r["Grid_Size_Y"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, []); a.append((int(r["Start_Timestamp"]), d))
for key, v in agg.items():
    ds = sorted(d for _, d in v)
    print(key, "launches", len(v), "median_us", round(ds[len(ds) // 2], 1), "min", round(ds[0], 1), "max", round(ds[-1], 1))
# gaps inside the last burst of Y > 1 launches (the one-call model pass)
multi = sorted((s, d) for k, v in agg.items() if int(k[1]) > 1 for s, d in v)
if multi:
    last = multi[-48:]
    span = (last[-1][0] + last[-1][1] * 1e3 - last[0][0]) / 1e3
    busy = sum(d for _, d in last)
    print("last", len(last), "multi-matrix launches: span_us", round(span), "kernel_sum_us", round(busy), "gaps_us", round(span - busy))
PY
