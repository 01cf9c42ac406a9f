#!/bin/bash
# Lab build: recompile ONE source of the library with extra -D switches and link it with the shipped objects of the others
# into build/lab/<name>.so (never the shipped library).   bash scripts/lab_build_variant.sh ring_stamps rtn.hip -DOQ_RING_STAMPS
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
python -m onnx_quantize_amd._build > /dev/null
mkdir -p "$root/build/lab"
extra=""
[ "$src" = "hqq.hip" ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $extra "$@" \
    -c "$root/onnx_quantize_amd/csrc/$src" -o "$root/build/lab/${name}_${src%.hip}.o"
objs=""
for o in "$root"/build/oq_hip/*.o; do
    [ "$(basename "$o")" = "${src%.hip}.o" ] || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/lab/$name.so" $objs "$root/build/lab/${name}_${src%.hip}.o"
echo "$root/build/lab/$name.so"
