#!/bin/bash
# Lab: rtn_resident_stream2 (communication wave) against rtn_resident_stream and the three-launch path: digests and times.
# $1 = library (a -DOQ_SPIN_LIMIT build for first runs).  Each variant in its own process (the choice is read once), each
# under its own time limit; a step that fails stops the script.
set -e
LIB=${1:-onnx_quantize_amd/lib/liboq_hip.so}
OUT=gpurun_out/lab_stream2
mkdir -p $OUT
SH="4096x11008,11008x4096,4096x4096,8192x8192,2048x8192,4100x1000"
OQ_RTN_RESIDENT=0 timeout -k 10 200 python scripts/quick_strategies.py --lib $LIB --reps 20 --shapes $SH --json $OUT/three.json > $OUT/three.log 2>&1
OQ_RTN_RES_TILE=2 timeout -k 10 200 python scripts/quick_strategies.py --lib $LIB --reps 100 --shapes $SH --json $OUT/s2.json > $OUT/s2.log 2>&1
OQ_RTN_RES_TILE=1 timeout -k 10 200 python scripts/quick_strategies.py --lib $LIB --reps 100 --shapes $SH --json $OUT/s1.json > $OUT/s1.log 2>&1
timeout -k 10 200 python scripts/quick_strategies.py --lib $LIB --reps 100 --shapes $SH --json $OUT/auto.json > $OUT/auto.log 2>&1
python - <<'PY'
import json
o="gpurun_out/lab_stream2/"
t,s2,s1,au=(json.load(open(o+f))["rows"] for f in ("three.json","s2.json","s1.json","auto.json"))
for a,b,c,d in zip(t,s2,s1,au):
    if a["strategy"]=="tensor": continue
    print(f'{a["shape"]:>11} {a["qtype"]:5} {a["strategy"]:7} g={a["g"]:<5} three {a["us"]:7.1f}  stream {c["us"]:7.1f}  stream2 {b["us"]:7.1f}  auto {d["us"]:7.1f}   digests {"OK" if a["digest"]==b["digest"]==c["digest"]==d["digest"] else "DIFFER "+a["digest"][:12]+" "+b["digest"][:12]+" "+c["digest"][:12]}')
PY
