#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*\|SQ_[A-Z_0-9]*ICACHE[A-Z_0-9]*\|SQC_[A-Z_0-9]*" | sort -u | head -40 > $OUT/counters.txt
cat $OUT/counters.txt | tr '\n' ' '
echo
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o t -- python3 $GRAFT_REPO_ROOT/scripts/quick_loop.py > $OUT/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o t -- python3 $GRAFT_REPO_ROOT/scripts/quick_loop.py > $OUT/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_run3"
for d in ("pmc1", "pmc2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "rows16" in r["Kernel_Name"] or "gemm_tn" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, "n=", len(next(iter(v.values()))))
PY
