#!/bin/bash
# Everything the round's documents quote, from ONE binary on ONE box:   gpurun --timeout 3000 -- 'bash tools/final_round_run.sh r4'
TAG=${1:-r4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -c 400 $OUT/bench.json; echo
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1
python bench.py --config depthformer_a.py --no-cpu-baseline --no-fp32 > $OUT/bench_config3.json 2> /dev/null
python bench.py --config depthformer_a_ddad.py --height 1216 --width 1936 --batch 1 --no-cpu-baseline --no-fp32 > $OUT/bench_config4.json 2> /dev/null
python bench.py --attn fp8 --no-cpu-baseline --no-fp32 > $OUT/bench_config5.json 2> /dev/null
python tools/library_roofline.py > $OUT/library_roofline.txt 2>&1
python tools/ubench/aten_sites.py > $OUT/aten_call_sites.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1
tail -4 $OUT/gputest.log
cp gpurun_out/parity_e2e.json $OUT/parity_e2e.json 2>/dev/null
for f in $OUT/bench_config*.json; do head -c 250 $f; echo; done
