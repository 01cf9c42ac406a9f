#!/bin/bash
# Round 6: cache policy of the quantized bytes' stores in the ticketed kernels (build/lab/res_store{1..5}.so = rtn_resident.hip with
# -DOQ_RES_STORE=n; the shipped library = 0): per-tensor and per-channel calls, us per call and digests.  gpurun_out/store_policy.txt
set -e
O=gpurun_out/store_policy.txt
: > $O
SHAPES=${SHAPES:-4096x11008,8192x8192,4096x4096}
for rnd in 1 2; do for v in 0 1 2 3 4 5; do
  lib=""; [ $v != 0 ] && lib="--lib build/lab/res_store$v.so"
  for st in tensor channel; do
    echo "== store policy $v, $st (round $rnd)" >> $O
    timeout -k 10 240 python scripts/quick_strategies.py $lib --shapes "$SHAPES" --only $st 2>&1 | grep '"us"' | grep -v uint4 | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ', r['shape'], r['strategy'], r['us'], r['digest'][:8])" >> $O
  done
done; done
cat $O
