#!/bin/bash
# Build and run the streaming-RTN experiment harness on the GPU box:  gpurun -- bash scripts/run_stream_bench.sh [N] [reps]
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o gpurun_out/rtn_stream_bench scripts/rtn_stream_bench.hip \
      -Lonnx_quantize_amd/lib -loq_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/onnx_quantize_amd/lib"
timeout -k 10 240 ./gpurun_out/rtn_stream_bench "$@" | tee gpurun_out/stream_bench_${TAG:-run}.log
