#!/bin/bash
# Block-order sweep of the group RTN kernels over widths and layouts (round 5: what the width rules in rtn.hip's dispatch are
# taken from).  One process per setting (scripts/lab_knob_sweep.sh -> scripts/lab_rtn_shapes.py); tables on stdout.
#   bash scripts/lab_order_sweep.sh out_dir
out=$1; mkdir -p "$out"
SH=${SHAPES:-4096x2048,4096x4096,4096x5120,4096x8192,4096x11008,4096x13824,4096x14336,4096x16384,4096x28672,4096x32000,11008x4096,14336x4096,8192x8192}
here=$(dirname "$0")
echo "## nbits"; bash "$here/lab_knob_sweep.sh" "$out/nbits.log" nbits "$SH" "XG=0" "XG=1" "XG=2" "XG=3" || exit 1
for lay in kn kn_packed4; do
    echo "## $lay"; bash "$here/lab_knob_sweep.sh" "$out/$lay.log" $lay "$SH" "ORDER=1" "ORDER=2 XG=0" "ORDER=2 XG=1" "ORDER=2 XG=2" "ORDER=2 XG=3" || exit 1
done
