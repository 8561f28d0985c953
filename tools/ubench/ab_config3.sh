cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab3
ARGS="--config depthformer_a.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing --steps 30 --warmup 8"
python bench.py $ARGS > /dev/null 2>&1
for off in none gemm msda_mm msda_value_raw conv3x3_wgrad gemm,msda_mm,msda_value_raw,conv3x3_wgrad none; do
  v=$([ "$off" = none ] && echo "" || echo "$off")
  ms=$(GE_DISABLE="$v" python bench.py $ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "GE_DISABLE=$off  $ms ms/step"
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab3/stats -- python $R/bench.py --config depthformer_a.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing --steps 20 --warmup 5 > $R/gpurun_out/ab3/stats_bench.json 2> /dev/null
cp $(find $R/gpurun_out/ab3/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/ab3/kernel_stats.csv
rm -rf $R/gpurun_out/ab3/stats
head -c 300 $R/gpurun_out/ab3/stats_bench.json
