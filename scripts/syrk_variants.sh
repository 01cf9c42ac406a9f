#!/bin/bash
# A/B of SYRK kernel builds on one box: each variant library replaces the in-tree one (scratch copy of the repo) for one run
cd $GRAFT_REPO_ROOT
O=gpurun_out/syrk_variants.log; mkdir -p gpurun_out; : > $O
cp onnx_quantize_amd/lib/liboq_hip.so /tmp/liboq_orig.so
for rep in 1 2; do
for v in "$@"; do
  cp build/variants/liboq_$v.so onnx_quantize_amd/lib/liboq_hip.so
  echo "== $v (rep $rep)" >> $O
  timeout -k 10 200 python3 scripts/quick_hessian.py f16x3 >> $O 2>&1 || { cp /tmp/liboq_orig.so onnx_quantize_amd/lib/liboq_hip.so; tail -5 $O; exit 1; }
done
done
cp /tmp/liboq_orig.so onnx_quantize_amd/lib/liboq_hip.so
grep -v amdgpu.ids $O
