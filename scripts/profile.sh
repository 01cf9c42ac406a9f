#!/bin/bash
# The round's profiles (run on the GPU box through gpurun): every summary that profiles/rNN_* is made of (rounds 4 and 5; it was
# scripts/profile_r04.sh).  Kernel traces (--kernel-trace --stats) and PMC passes are separate runs; the program comes
# directly after `--`.
#   ROUND=r05 scripts/profile.sh [rtn] [strategies] [packed] [shapes] [searches] [awq] [gptq] [file]      (default: all but file)
set -u
ROUND=${ROUND:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WHAT="${*:-rtn strategies packed shapes searches awq gptq}"
trace() { rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1/trace -- "${@:2}" > $OUT/$1.trace.log 2>&1; }
pmc() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1/pmc_$2 -- "${@:3}" > $OUT/$1.$2.log 2>&1; }
for w in $WHAT; do
  case $w in
    rtn)
      for lay in nbits kn; do
        B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --layout $lay"
        trace rtn_$lay $B; pmc rtn_$lay FETCH_SIZE $B; pmc rtn_$lay WRITE_SIZE $B
      done;;
    strategies)
      P="python3 $R/scripts/quick_strategies.py --reps 100 --shapes 4096x11008,11008x4096"
      trace strategies $P; pmc strategies FETCH_SIZE $P; pmc strategies WRITE_SIZE $P;;
    packed)
      P="python3 $R/scripts/lab_rtn_shapes.py --layout kn_packed4 --qtype int4 --shapes 4096x11008 --reps 100 --trials 1"
      trace packed $P; pmc packed FETCH_SIZE $P; pmc packed WRITE_SIZE $P;;
    shapes)     # the widths whose block order changed in round 5, blob and [K,N] bytes
      P="python3 $R/scripts/lab_rtn_shapes.py --layout nbits --shapes 4096x4096,11008x4096,8192x8192,4096x32000 --reps 100 --trials 1"
      trace shapes_nbits $P
      P="python3 $R/scripts/lab_rtn_shapes.py --layout kn --shapes 4096x4096,11008x4096 --reps 100 --trials 1"
      trace shapes_kn $P;;
    searches)
      P="python3 $R/scripts/quick_searches.py"
      trace searches $P;;
    awq)
      P="python3 $R/scripts/quick_awq.py"
      trace awq $P;;
    gptq)
      trace gptq python3 $R/bench_gptq.py --layers 8 --no-cpu-baseline --hessian-methods "" --extra-passes corrected;;
    pmc)        # the launches whose roofline.traffic had no counter record (calibration minmax, Hessian SYRK): round 6
      for t in calibration hessian; do
        P="python3 $R/scripts/pmc_targets.py $t"
        trace pmc_$t $P; pmc pmc_$t FETCH_SIZE $P; pmc pmc_$t WRITE_SIZE $P
      done;;
    file)       # the file path: calibration walk on torch-ROCm + own GEMM, Hessians, factors, loops, packs
      trace file_gptq python3 $R/bench_model.py --layers 4 --config gptq_int4_g128 --samples 16 --seq 2048 --repeat 1
      trace file_awq python3 $R/bench_model.py --layers 4 --config awq_uint4_g128 --samples 16 --seq 2048 --repeat 1;;
  esac
  python3 $R/scripts/summarize_kernels.py $OUT/$w* > /dev/null 2>&1
  echo "$w done"
done
for d in $OUT/*/; do n=$(basename $d); python3 $R/scripts/summarize_kernels.py $d > $OUT/$n.summary.json 2> $OUT/$n.summary.err; done
ls $OUT/*.summary.json
