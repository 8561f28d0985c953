#!/bin/bash
# Everything the round's documents quote, from ONE binary on ONE box:   gpurun --timeout 5400 -- 'bash tools/final_round_run.sh r6'
TAG=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $REPO
python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -c 400 $OUT/bench.json; echo
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1
python bench.py --config depthformer_a.py --no-cpu-baseline --no-fp32 > $OUT/bench_config3.json 2> /dev/null                       # hipGraph (auto: 2 images per GPU)
python bench.py --config depthformer_a.py --no-cpu-baseline --no-fp32 --graph off > $OUT/bench_config3_eager.json 2> /dev/null
python bench.py --config depthformer_a_ddad.py --height 1216 --width 1936 --batch 1 --no-cpu-baseline --no-fp32 > $OUT/bench_config4.json 2> /dev/null
# (config #5, --attn fp8, is a numerics demonstrator since round 6: tested, not benched — DESIGN.md §8.2)
# gradient exchange forced on one rank: bucket layout in arrival order + per-bucket launch / completion trace
GE_DDP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing > $OUT/bench_ddp_forced.json 2> /dev/null
GE_DDP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --config depthformer_a.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing > $OUT/bench_ddp_forced_config3.json 2> /dev/null
GE_GRAPH_DDP=0 GE_DDP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --config depthformer_a.py --no-cpu-baseline --no-fp32 --no-h2d --no-kernel-timing > $OUT/bench_ddp_forced_config3_eager.json 2> /dev/null
python tools/library_roofline.py > $OUT/library_roofline.txt 2>&1
python tools/ubench/aten_sites.py > $OUT/aten_call_sites.txt 2>&1
CFG=depthformer_a.py python tools/ubench/aten_sites.py > $OUT/aten_call_sites_config3.txt 2>&1
for g in model concentrated spread; do python tools/ubench/msda_mm/dv_time.py $g 2>&1 | grep -v amdgpu.ids; done > $OUT/dv_time.txt
# round 6: d_value on the value-stationary kernel vs the record pipeline; the self-attention's level split; the cross-attention's query order
for g in model self random; do python tools/ubench/msda_mm/vs_time.py $g 2>&1 | grep -v amdgpu.ids; done > $OUT/dv_value_stationary.txt
python tools/ubench/msda_mm/self_split_time.py 2>&1 | grep -v amdgpu.ids > $OUT/self_split_time.txt
python tools/ubench/msda_mm/order_time.py 2>&1 | grep -v amdgpu.ids > $OUT/order_time.txt
python tools/ubench/winattn_time.py 2>&1 | grep -v amdgpu.ids > $OUT/winattn_time.txt
# config 3: GPU idle share with and without the hipGraph (kernel trace of the last 10 steps)
( cd /tmp && export TMPDIR=/tmp; A="--no-cpu-baseline --no-fp32 --no-kernel-timing --no-h2d --config depthformer_a.py --steps 20 --warmup 5"
  for g in on off; do rm -rf /tmp/c3$g; rocprofv3 --kernel-trace --output-format csv -d /tmp/c3$g -- python $REPO/bench.py $A --graph $g > /dev/null 2>&1
    echo "graph $g"; python $REPO/tools/ubench/graph/busy.py $(find /tmp/c3$g -name "*kernel_trace.csv" | head -1) 10 30; done ) > $OUT/config3_busy.txt 2>&1
# the GPU suite THREE times in a row on this binary, as the driver runs it (-x): a flaky test must show here, not in the driver's run
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputest_$i.log 2>&1; tail -1 $OUT/gputest_$i.log; done
cp $OUT/gputest_3.log $OUT/gputest.log
cp gpurun_out/parity_e2e.json $OUT/parity_e2e.json 2>/dev/null
for f in $OUT/bench_config*.json $OUT/bench_ddp*.json; do head -c 250 $f; echo; done
