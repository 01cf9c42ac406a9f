#!/bin/bash
# build a lab copy of the library with the slice-count knob, run the sweep, rebuild the shipped library
set -e
python3 -m onnx_quantize_amd._build --define OQ_SYRK_LAB > /dev/null
python3 scripts/lab_syrk_splits.py
python3 -m onnx_quantize_amd._build > /dev/null
