#!/usr/bin/env python
"""Regenerate the committed library-selection tables (gedepth_amd/tuning/) on an MI355X:

  * ``tunableop_gfx950.csv`` — PyTorch TunableOp winners for the GEMM / batched-GEMM shapes of a workload,
  * ``miopen/*.u{f,}db.txt`` — MIOpen's find-db after an exhaustive find of the workload's convolutions.

    python tools/tune_tables.py --config depthformer_swint_v.py                       # the bench workload (8 x 352 x 1120)
    python tools/tune_tables.py --config depthformer_a.py --batch 2                   # Swin-L + GEDepth-Adaptive
    python tools/tune_tables.py --dtype fp32                                          # the fp32 problems of the bench workload

Existing entries are kept (both tables are seeded from the committed files and appended to); takes ~5-8 minutes per
workload, almost all of it MIOpen's solver timing.  Nothing here runs during training or benchmarking: those only look
the tables up (gedepth_amd/mmrt/tuning.py).
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNING = os.path.join(ROOT, 'gedepth_amd', 'tuning')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='depthformer_swint_v.py')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--height', type=int, default=352)
    ap.add_argument('--width', type=int, default=1120)
    ap.add_argument('--layout', default='nhwc', choices=['nchw', 'nhwc'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'], help='fp32: the convolutions / GEMMs of the reference-precision step (the drop-in default)')
    ap.add_argument('--out-dir', default=None, help='also copy the updated tables here (e.g. gpurun_out/tuning on the GPU box)')
    a = ap.parse_args()
    work = tempfile.mkdtemp(prefix='gedepth_tune_')
    db = os.path.join(work, 'miopen')
    os.makedirs(db)
    for f in os.listdir(os.path.join(TUNING, 'miopen')):
        shutil.copy(os.path.join(TUNING, 'miopen', f), db)
    table = os.path.join(work, 'tunableop.csv')
    shutil.copy(os.path.join(TUNING, 'tunableop_gfx950.csv'), table)
    env = dict(os.environ, MIOPEN_USER_DB_PATH=db, GE_GEMM_TABLE=table, PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS='30',
               PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS='5')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', a.config, '--steps', '3', '--warmup', '3', '--no-cpu-baseline',
           '--no-kernel-timing', '--cudnn-benchmark', '1', '--gemm-tuning', 'tune', '--height', str(a.height), '--width', str(a.width), '--layout', a.layout,
           '--no-fp32', '--dtype', a.dtype, '--no-h2d']
    if a.batch:
        cmd += ['--batch', str(a.batch)]
    subprocess.run(cmd, env=env, check=True)
    shutil.copy(table, os.path.join(TUNING, 'tunableop_gfx950.csv'))
    for f in os.listdir(db):
        if f.endswith('db.txt'):
            shutil.copy(os.path.join(db, f), os.path.join(TUNING, 'miopen', f))
    if a.out_dir:
        os.makedirs(os.path.join(a.out_dir, 'miopen'), exist_ok=True)
        shutil.copy(os.path.join(TUNING, 'tunableop_gfx950.csv'), a.out_dir)
        for f in os.listdir(os.path.join(TUNING, 'miopen')):
            shutil.copy(os.path.join(TUNING, 'miopen', f), os.path.join(a.out_dir, 'miopen'))
    print(f'updated {TUNING}')


if __name__ == '__main__':
    main()
