#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun): every summary that profiles/r03_* is made of.
#   kernel traces (--kernel-trace --stats) and PMC passes are separate runs; the program comes directly after `--`.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10"
# 1. headline RTN kernel, both layouts: kernel trace + HBM traffic counters
for lay in nbits kn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rtn_$lay/trace -- $B --layout $lay > $OUT/rtn_$lay.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rtn_$lay/pmc_fetch -- $B --layout $lay > $OUT/rtn_$lay.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rtn_$lay/pmc_write -- $B --layout $lay > $OUT/rtn_$lay.write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/rtn_$lay/pmc_sq -- $B --layout $lay > $OUT/rtn_$lay.sq.log 2>&1
  echo "rtn $lay done"
done
# 2. GPTQ, 8 of 32 layers, parity pass + corrected pass in one process: kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gptq/trace -- python3 $R/bench_gptq.py --layers 8 --no-cpu-baseline --hessian-methods "" --extra-passes corrected > $OUT/gptq.trace.log 2>&1
echo "gptq trace done"
# 3. the corrected loop alone (three Llama shapes) and the Hessian alone: kernel trace + SQ / MFMA counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop/trace -- python3 $R/scripts/quick_loop.py > $OUT/loop.trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/loop/pmc_sq -- python3 $R/scripts/quick_loop.py > $OUT/loop.sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/loop/pmc_mfma -- python3 $R/scripts/quick_loop.py > $OUT/loop.mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hess/trace -- python3 $R/scripts/quick_hessian.py f16x3 11008 > $OUT/hess.trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/hess/pmc_mfma -- python3 $R/scripts/quick_hessian.py f16x3 11008 > $OUT/hess.mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/factor/trace -- python3 $R/scripts/lab_factor_trace.py 11008 8 > $OUT/factor.trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hess_many/trace -- python3 $R/scripts/quick_hessian_many.py > $OUT/hess_many.trace.log 2>&1
echo "loop / hessian / factor done"
# 4. calibration (config 3 stand-in) and the AWQ searches: kernel traces
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/calib/trace -- python3 $R/bench_calib.py --no-cpu-baseline > $OUT/calib.trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/awq/trace -- python3 $R/scripts/quick_awq.py > $OUT/awq.trace.log 2>&1
echo "calib / awq done"
python3 $R/scripts/summarize_r03.py $OUT > $OUT/summary.log 2>&1; tail -3 $OUT/summary.log
