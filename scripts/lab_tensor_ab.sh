#!/bin/bash
# Lab: two builds of the per-tensor kernel on one box, alternating (times), then the phase stamps of each.  $1, $2 = names under build/lab/
set -e
O=gpurun_out/lab_tensor_ab.txt
: > $O
for rnd in 1 2; do for n in $1 $2; do
  echo "== $n (round $rnd)" >> $O
  timeout -k 10 120 python scripts/quick_strategies.py --lib build/lab/$n.so --reps 200 --shapes 4096x11008,11008x4096,8192x8192 2>&1 | grep tensor | cut -c1-120 >> $O
done; done
for n in $1 $2; do
  echo "== stamps $n" >> $O
  timeout -k 10 120 python scripts/lab_tensor_stamps.py build/lab/${n}_stamps.so 2>&1 | grep -v amdgpu >> $O
done
cat $O
