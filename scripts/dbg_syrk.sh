#!/bin/bash
# timing-only experiments on the split-operand SYRK: rebuild the one object with -DOQ_SYRK_DBG=<mask> and time it
cd $GRAFT_REPO_ROOT
for d in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DOQ_SYRK_DBG=$d -c onnx_quantize_amd/csrc/syrk_bf16x3.hip -o build/oq_hip/syrk_bf16x3.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o onnx_quantize_amd/lib/liboq_hip.so build/oq_hip/*.o || exit 1
  echo "== DBG=$d"
  OQ_HESSIAN_METHOD=2 python3 scripts/quick_hess_time.py 2>&1 | grep "K=11008 T=65536\|K=4096 T=65536"
done
