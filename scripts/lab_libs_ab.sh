#!/bin/bash
# Lab: several builds of the library on one box, alternating: quick_strategies.py rows (us per call, digest).
#   STRATEGY=channel SHAPES=4096x11008,8192x8192 bash scripts/lab_libs_ab.sh build/lab/a.so build/lab/b.so ...   -> gpurun_out/libs_ab.txt
set -e
mkdir -p gpurun_out
O=gpurun_out/libs_ab.txt
: > $O
SHAPES=${SHAPES:-4096x11008,11008x4096,8192x8192,4096x4096}
for rnd in 1 2; do for lib in "$@"; do
  echo "== $lib (round $rnd)" >> $O
  timeout -k 10 240 python scripts/quick_strategies.py --lib $lib --shapes "$SHAPES" --only ${STRATEGY:-channel} 2>&1 | grep '"us"' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('  ', r['shape'], r['qtype'], r['strategy'], r['g'], r['us'], r['digest'][:26])" >> $O
done; done
cat $O
