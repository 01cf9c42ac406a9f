#!/bin/bash
# Lab: what bounds the streamed per-channel kernel -- the hand-off or the order in which the matrix is read?
set -e
export OQ_RTN_RES_TILE=1
O=gpurun_out/lab_order.txt
: > $O
for n in nh nhcf nhcf4 nhcf8 cf2 cf4; do
  echo "== $n" >> $O
  timeout -k 10 120 python scripts/quick_strategies.py --lib build/lab/$n.so --reps 100 --shapes ${1:-4096x11008,8192x8192} 2>&1 | grep -E "channel|spin" | grep -E "int8|spin" >> $O
done
echo "== shipped" >> $O
timeout -k 10 120 python scripts/quick_strategies.py --reps 100 --shapes ${1:-4096x11008,8192x8192} 2>&1 | grep -E "channel" | grep int8 >> $O
cut -c1-140 $O
