#!/bin/bash
# rocprofv3 of per-channel / per-tensor RTN on 4096 x 11008 int8: kernel trace and the HBM traffic
# counters in SEPARATE runs, the program directly after `--`.  $1 = tag (old | new); "old" exports OQ_RTN_RESIDENT=0
# (the three-launch path that reads W twice) before rocprofv3 starts -- an exported variable, not an `env` hop.
set -u
TAG=${1:-new}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_strategies_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$TAG" = "old" ]; then export OQ_RTN_RESIDENT=0; fi
P="python3 $GRAFT_REPO_ROOT/scripts/quick_strategies.py --reps 100 --shapes 4096x11008"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $P > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $P > $OUT/write.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/summarize_kernels.py $OUT > $OUT/summary.json 2> $OUT/summary.err
tail -5 $OUT/trace.log
