cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/awq
for v in fused unfused; do
  lib=$R/onnx_quantize_amd/lib/liboq_hip.so; [ $v = unfused ] && lib=$R/build/lab/awq_unfused.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/awq/prof_$v -- python3 $R/scripts/lab_awq_loss_forms.py $lib $R/gpurun_out/awq/x_$v.pt > $R/gpurun_out/awq/prof_$v.log 2>&1
  f=$(find $R/gpurun_out/awq/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -12 "$f" | cut -d, -f1-4 | cut -c1-150
done
