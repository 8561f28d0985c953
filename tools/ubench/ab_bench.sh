#!/bin/bash
# Same-box A/B of the round-3 fused passes: bench.py with everything on, then with one feature at a time switched to its two-pass
# composition (GE_DISABLE, gedepth_amd/kernels.py).  Box-to-box spread of the same binary is 2-4 %, so only these same-session numbers
# are quoted for small deltas.       gpurun -- 'bash tools/ubench/ab_bench.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing --steps 30 --warmup 8"
python bench.py $ARGS > /dev/null 2>&1          # warm caches
for round in 1 2; do
for off in ${AB_LIST:-none ln_res conv_lib conv3x3_c1 conv3x3_wgrad conv3x3 upcat upsum bias_gelu msda_raw conv3x3,conv3x3_c1,conv3x3_wgrad,upcat,upsum,bias_gelu,msda_raw}; do
  v=$([ "$off" = none ] && echo "" || echo "$off")
  ms=$(GE_DISABLE="$v" python bench.py $ARGS 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "round $round  GE_DISABLE=$off  $ms ms/step"
done
done
