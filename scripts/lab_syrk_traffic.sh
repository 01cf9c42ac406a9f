#!/bin/bash
# HBM and L2 traffic of the fp16-piece product (K = 11008 and 4096, 65 536 rows): separate PMC passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/syrk_traffic; mkdir -p $OUT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -- python3 $R/scripts/quick_hessian.py f16x3 4096,11008 > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/syrk_traffic"
for d in sorted(os.listdir(root)):
    fs = glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "syrk_f16_m16" in r["Kernel_Name"]:
            agg[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (name, grid), v in sorted(agg.items()):
        v = sorted(v)
        print(d, name, "grid", grid, "launches", len(v), "median", v[len(v) // 2])
PY
