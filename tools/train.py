#!/usr/bin/env python
"""Train a depther — same CLI as the reference's tools/train.py:22-63 (config, --work-dir, --load-from, --resume-from,
--no-validate, --seed, --options k=v, --launcher), on the MI355X runtime.

    python tools/train.py configs/depthformer/depthformer_swint_v.py --synthetic 64 --options runner.max_iters=20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch

Without ``--synthetic`` the KITTI tree named by ``cfg.data`` is used (gedepth_amd/depth/datasets/kitti.py, SURVEY.md §8
f1); ``--synthetic N`` trains on N seeded KITTI-shaped samples instead (gedepth_amd/depth/datasets/synthetic.py).
"""
import argparse
import os
import os.path as osp
import sys
import time

# ROCm 7.2: graph replays that run while another stream copies host -> device faulted ("illegal memory access") with the runtime's
# AQL-packet capture of graph kernels on; without it the same runs are clean and as fast (gedepth_amd/mmrt/graph.py).  Read at HIP start-up.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch  # noqa: E402

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

from gedepth_amd import __version__  # noqa: E402
from gedepth_amd.depth.apis.train import set_random_seed, train_depther  # noqa: E402
from gedepth_amd.depth.datasets.loader import SyntheticKITTI  # noqa: E402
from gedepth_amd.depth.models import build_depther  # noqa: E402
from gedepth_amd.mmrt.config import Config, DictAction  # noqa: E402
from gedepth_amd.mmrt.ddp import init_dist  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Train a depther')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work-dir', help='the dir to save logs and models')
    p.add_argument('--load-from', help='the checkpoint file to load weights from')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--no-validate', action='store_true')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--options', nargs='+', default=None, help='override settings: k=v pairs')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--synthetic', type=int, default=0, help='train on N synthetic KITTI-shaped samples')
    p.add_argument('--bf16', action='store_true', help='bf16 autocast (fp32 master weights); default: the config\'s `amp`, else fp32 like the reference')
    p.add_argument('--fp32', action='store_true', help='force fp32 even if the config sets amp = "bf16"')
    p.add_argument('--gpu-pipeline', action='store_true',
                   help='KITTI: loader workers only decode files; ground depth from the calibration and every transform of the train '
                        'pipeline run on the GPU (gedepth_amd/depth/datasets/gpu_pipeline.py, SURVEY.md §8 f3)')
    p.add_argument('--pe-source', default='calib', choices=['calib', 'npy'], help='--gpu-pipeline: ground depth from the calibration files or from pe_165.npy')
    p.add_argument('--hip-graph', action='store_true',
                   help='capture the training step in a hipGraph after three eager iterations (gedepth_amd/mmrt/graph.py): one launch per iteration')
    p.add_argument('--layout', default='nhwc', choices=['nchw', 'nhwc'], help='nhwc: channels-last conv stack (depth/models/utils/layout.py)')
    p.add_argument('--gemm-tuning', default='load', choices=['off', 'load', 'tune'],
                   help='hipBLASLt/rocBLAS solution table for the Linear layers (gedepth_amd/mmrt/tuning.py)')
    return p.parse_args()


def main():
    args = parse_args()
    cfg = Config.fromfile(args.config)
    if args.options:
        cfg.merge_from_dict(DictAction.parse(args.options))
    cfg.work_dir = args.work_dir or cfg.get('work_dir') or osp.join('./work_dirs', osp.splitext(osp.basename(args.config))[0])
    if args.load_from:
        cfg.load_from = args.load_from
    if args.resume_from:
        cfg.resume_from = args.resume_from
    if args.hip_graph:
        cfg.hip_graph = True
    if args.bf16:
        cfg.amp = 'bf16'
    if args.fp32:
        cfg.amp = 'fp32'
    distributed = args.launcher != 'none'
    rank, local, world = init_dist(cfg.get('dist_params', {}).get('backend', 'nccl')) if distributed else (0, 0, 1)
    if not distributed:
        torch.cuda.set_device(0)
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    use_tuned_gemms(args.gemm_tuning)
    if cfg.get('cudnn_benchmark', False):
        use_miopen_find_db()
        torch.backends.cudnn.benchmark = True
    os.makedirs(cfg.work_dir, exist_ok=True)
    if rank == 0:
        cfg.dump(osp.join(cfg.work_dir, osp.basename(args.config)))
    timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
    if args.seed is not None:
        set_random_seed(args.seed + rank, deterministic=args.deterministic)
    cfg.seed = args.seed

    pretrained = cfg.model.get('pretrained')
    if pretrained and not osp.isfile(pretrained):
        print(f'[train] pretrained weights {pretrained} not found: training from random init')
        cfg.model.pretrained = None
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    evaluate_fn = data_loaders = batch_transform = None
    if args.synthetic > 0:
        h, w = cfg.get('crop_size', (352, 1120))
        dataset = SyntheticKITTI(args.synthetic, h, w, adaptive='dynamic_pe_neck' in cfg.model, seed=1234)
    else:                                                     # the KITTI tree of cfg.data (depth/datasets/kitti.py)
        from gedepth_amd.depth.apis.test import multi_gpu_test
        from gedepth_amd.depth.datasets import build_dataloader, build_dataset
        if args.gpu_pipeline:
            from gedepth_amd.depth.datasets.gpu_pipeline import KITTIGPUPipeline, KITTIRawDataset, raw_collate
            t = cfg.data.train
            dataset = KITTIRawDataset(img_dir=t.img_dir, ann_dir=t.ann_dir, split=t.split, data_root=t.get('data_root'),
                                      depth_scale=t.get('depth_scale', 256))
            sampler = torch.utils.data.DistributedSampler(dataset, world, rank, shuffle=True, seed=args.seed or 0) if distributed else None
            data_loaders = [torch.utils.data.DataLoader(dataset, batch_size=cfg.data.samples_per_gpu, sampler=sampler, shuffle=sampler is None,
                                                        num_workers=cfg.data.workers_per_gpu, collate_fn=raw_collate, drop_last=True,
                                                        persistent_workers=cfg.data.workers_per_gpu > 0)]
            batch_transform = KITTIGPUPipeline(data_root=t.get('data_root'), img_dir=t.img_dir, pe_source=args.pe_source,
                                               depth_scale=t.get('depth_scale', 256)).batch
        else:
            dataset = build_dataset(cfg.data.train)
        if not args.no_validate:
            val_set = build_dataset(cfg.data.val, dict(test_mode=True))
            val_loader = build_dataloader(val_set, 1, cfg.data.workers_per_gpu, dist=distributed, shuffle=False)

            def evaluate_fn(runner):
                res = multi_gpu_test(runner.model, val_loader, pre_eval=True)
                runner.model.train()
                return val_set.evaluate(res, logger=None) if res is not None else None
    meta = dict(gedepth_amd_version=__version__, config=cfg.pretty_text, seed=args.seed)
    log = (lambda m: print(m, flush=True)) if rank == 0 else (lambda m: None)
    train_depther(model, dataset, cfg, distributed=distributed, validate=not args.no_validate, timestamp=timestamp,
                  meta=meta, logger=log, evaluate_fn=evaluate_fn, data_loaders=data_loaders, batch_transform=batch_transform,
                  channels_last=args.layout == 'nhwc')


if __name__ == '__main__':
    main()
