#!/bin/bash
# SURVEY §8(f3) "done" criterion: tools/train.py on a KITTI-shaped tree with 2 loader workers per GPU — iteration time of the
# device data pipeline (--gpu-pipeline) vs the reference's host pipeline, next to bench.py's resident-batch step time.
#   gpurun -- 'bash tools/ubench/train_rate.sh'        -> gpurun_out/train_rate_{gpu,host}.log
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, 'tests')
from toy_kitti import make_toy_kitti
make_toy_kitti('/tmp/toy_kitti', frames=48)          # 96 raw-size frames (375 x 1242) + 1 without ground truth
PY
OPT="data.train.data_root=/tmp/toy_kitti data.train.split=/tmp/toy_kitti/split.txt data.workers_per_gpu=${WORKERS:-2} runner.max_iters=${ITERS:-50} log_config.interval=10 checkpoint_config.interval=100000"
python tools/train.py configs/depthformer/depthformer_swint_v.py --no-validate --seed 0 --work-dir /tmp/wd_gpu --gpu-pipeline --pe-source npy --options $OPT > gpurun_out/train_rate_gpu.log 2>&1
grep -E "Iter|iter|time" gpurun_out/train_rate_gpu.log | tail -4
python tools/train.py configs/depthformer/depthformer_swint_v.py --no-validate --seed 0 --work-dir /tmp/wd_host --options $OPT > gpurun_out/train_rate_host.log 2>&1
grep -E "Iter|iter|time" gpurun_out/train_rate_host.log | tail -4
