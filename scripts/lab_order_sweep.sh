#!/bin/bash
# Block-order sweep of the group RTN kernels over shapes and layouts (one process per setting; scripts/lab_rtn_shapes.py).
#   bash scripts/lab_order_sweep.sh out.log
out=$1
SH=${SHAPES:-4096x2048,4096x4096,4096x5120,4096x8192,4096x11008,4096x13824,4096x14336,4096x16384,4096x28672,4096x32000,11008x4096,14336x4096,8192x8192}
: > "$out"
run() {  # layout, settings
    envs=""; for kv in $2; do envs="$envs OQ_RTN_$kv"; done
    echo "== $1 $2" >> "$out"
    env $envs timeout -k 10 300 python scripts/lab_rtn_shapes.py --layout "$1" --reps 300 --trials 2 --shapes "$SH" >> "$out" 2>/dev/null || exit 1
    echo "done $1 $2"
}
for s in "XG=0" "XG=1" "XG=2" "XG=3"; do run nbits "$s"; done
for lay in kn kn_packed4; do
    for s in "ORDER=1" "ORDER=2 XG=0" "ORDER=2 XG=1" "ORDER=2 XG=2" "ORDER=2 XG=3"; do run $lay "$s"; done
done
