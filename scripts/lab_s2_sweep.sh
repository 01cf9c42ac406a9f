#!/bin/bash
# Lab: stream2 poll-sleep variants: laps of the stamp builds, times of the plain builds (4096x11008 and 11008x4096, int8 per channel).
set -e
export OQ_RTN_RES_TILE=2
O=gpurun_out/lab_s2_sweep.txt
: > $O
for n in a b c d; do
  echo "== stamps s2$n" >> $O
  timeout -k 10 120 python scripts/lab_stream2_laps.py build/lab/s2$n.so 4096x11008 >> $O 2>&1
  timeout -k 10 120 python scripts/lab_stream2_laps.py build/lab/s2$n.so 11008x4096 >> $O 2>&1
done
for n in a b c d e; do
  echo "== times s2t$n" >> $O
  timeout -k 10 120 python scripts/quick_strategies.py --lib build/lab/s2t$n.so --reps 100 --shapes 4096x11008,11008x4096,8192x8192 2>&1 | grep channel | grep int8 >> $O
done
unset OQ_RTN_RES_TILE
echo "== old stream (RES_TILE=1)" >> $O
OQ_RTN_RES_TILE=1 timeout -k 10 120 python scripts/quick_strategies.py --lib build/lab/s2ta.so --reps 100 --shapes 4096x11008,11008x4096,8192x8192 2>&1 | grep channel | grep int8 >> $O
cat $O | grep -v amdgpu.ids
