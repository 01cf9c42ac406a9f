#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
python -m onnx_quantize_amd._build --define OQ_LOOP_STAMPS > /dev/null 2>&1; echo "build rc=$?"
timeout -k 10 300 python scripts/lab_loop_stamps.py 2>&1 | grep -v amdgpu.ids
