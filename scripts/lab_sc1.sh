#!/bin/bash
# Round 6: write-through (`sc1`) stores of the quantized output in the group kernels (build/lab/rtn_sc1.so = rtn.hip with -DOQ_RTN_SC1=7)
# against the shipped library, same box, alternating.  gpurun_out/sc1.txt
set -e
O=gpurun_out/sc1.txt
: > $O
for rnd in 1 2; do for lib in "" "--lib build/lab/rtn_sc1.so"; do for lay in nbits kn kn_packed4; do
  echo "== ${lib:-shipped} $lay (round $rnd)" >> $O
  timeout -k 10 240 python scripts/lab_rtn_shapes.py $lib --layout $lay --shapes 4096x11008,11008x4096,4096x4096 2>&1 | grep '"us"' | cut -c1-200 >> $O
done; done; done
cat $O
