#!/bin/bash
# Lab: builds of the library on one box, alternating: lab_rtn_shapes.py on the [K,N] layouts (us per launch, digests).
#   bash scripts/lab_layouts_ab.sh onnx_quantize_amd/lib/liboq_hip.so build/lab/x.so   -> gpurun_out/layouts_ab.txt
set -e
mkdir -p gpurun_out
O=gpurun_out/layouts_ab.txt
: > $O
for rnd in 1 2; do for lib in "$@"; do for lay in ${LAYOUTS:-kn kn_packed4}; do
  echo "== $lib $lay (round $rnd)" >> $O
  timeout -k 10 240 python scripts/lab_rtn_shapes.py --lib $lib --layout $lay --shapes ${SHAPES:-4096x11008,11008x4096,4096x4096} 2>&1 | grep '"us"' | cut -c1-160 >> $O
done; done; done
cat $O
