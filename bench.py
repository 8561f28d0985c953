#!/usr/bin/env python
"""Throughput of the GEDepth training hot path on MI355X (BASELINE.json metric: training img/s, KITTI 352x1120).

    python bench.py --gpus 1 --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank/GPU)

A step = forward (DepthFormer Swin encoder + HAHI neck + ground embedding + DenseDepth head) + SiLog(+CE) loss +
backward + gradient all-reduce (N>1) + grad-clip + AdamW on one synthetic KITTI-shaped batch that is already
resident in HBM.  Prints ONE JSON line (rank 0) with the whole-job img/s, the roofline of the dominant hand-written
kernel (HIP-event timed inside the timed region) and, at N=1, the CPU oracle timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='depthformer_swint_v.py', help='file under configs/depthformer/')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: data.samples_per_gpu)')
    ap.add_argument('--height', type=int, default=352)
    ap.add_argument('--width', type=int, default=1120)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--gemm-tuning', default='load', choices=['off', 'load', 'tune'],
                    help='hipBLASLt/rocBLAS solution table for the Linear layers (gedepth_amd/mmrt/tuning.py)')
    ap.add_argument('--cudnn-benchmark', type=int, default=-1,
                    help='MIOpen find mode (configs: cudnn_benchmark=True): +6 %% step rate; -1 = on when the committed find-db '
                         '(gedepth_amd/tuning/miopen) is usable, else off (finding from scratch takes ~5 min on a fresh box)')
    return ap.parse_args()


def cpu_baseline(cfg_name, H, W):
    """The CPU oracle (oracle/gedepth_oracle.py, a port validated against reference-generated fixtures) timed on the
    host cores: one full training step (fwd + losses + bwd + clip + AdamW) on ONE synthetic image."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from oracle import gedepth_oracle as O
    from oracle.fill import fill_state_dict
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    cores = min(os.cpu_count() or 1, 32)        # more threads only add synchronisation overhead for these op sizes
    torch.set_num_threads(cores)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', cfg_name))
    cfg.model.pretrained = None
    model = build_depther(cfg.model)                     # host-side: only used for names / shapes
    P = fill_state_dict(model.state_dict(), 'bench')
    del model
    arch = dict(embed_dims=cfg.model.backbone.embed_dims, depths=list(cfg.model.backbone.depths),
                num_heads=list(cfg.model.backbone.num_heads), adaptive='dynamic_pe_neck' in cfg.model)
    leaves = [v.requires_grad_(True) for k, v in P.items()
              if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))]
    opt = torch.optim.AdamW(leaves, lr=1e-4, weight_decay=0.01)
    b = synthetic_batch(1, H, W, seed=1234)
    t0 = time.perf_counter()
    losses, _ = O.forward_train(b['img'], b['depth_gt'], b['pe_k_gt'], P, arch, train_bn=True)
    loss, _ = O.parse_losses(losses)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(leaves, 35.0)
    opt.step()
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 5), unit='img/s', cores=cores, kind='port',
                sample=f'1 training step (fwd+loss+bwd+clip+AdamW) on 1 image {H}x{W}, fp32, torch {torch.__version__} CPU, '
                       f'{dt:.1f} s, no warm-up')


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same workload); None when not collected."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        table = json.load(f).get('kernels', {})
    for key, rec in table.items():
        if kernel.startswith(key) or key.startswith(kernel):
            return rec.get('hbm_bytes_per_launch')
    return None


def main():
    args = parse()
    from gedepth_amd.mmrt.ddp import FlatDDP, init_dist
    rank, local, world = init_dist('nccl')
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; there is no CPU fallback'
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    have_db = use_miopen_find_db() if args.cudnn_benchmark != 0 else False
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark == 1 or (args.cudnn_benchmark == -1 and have_db))
    use_tuned_gemms(args.gemm_tuning)

    from gedepth_amd import hip, kernels
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    hip.lib()

    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', args.config))
    cfg.model.pretrained = None                             # random-init weights of the named architecture
    per_gpu = args.batch or cfg.data.samples_per_gpu
    torch.manual_seed(1234)
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    model = model.to(dev).train()
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    ddp = FlatDDP(model, optimizer.arena)
    batch = synthetic_batch(per_gpu, args.height, args.width, seed=1234 + rank, device=dev)
    amp = args.dtype == 'bf16'

    def step():
        optimizer.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = ddp.train_step(batch, optimizer)
        out['loss'].backward()
        ddp.finish()
        optimizer.step()
        return out

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    if not args.no_kernel_timing:
        kernels.PROFILER.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernels.PROFILER.disable()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    loss = out['log_vars']['loss']

    if rank == 0:
        total_imgs = per_gpu * world * args.steps
        res = {
            'metric': 'training img/s (whole node), KITTI-shaped 352x1120 synthetic batches',
            'value': round(total_imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.config[:-3]}: DepthFormer-Swin{"T" if cfg.model.backbone.embed_dims == 96 else "L"} + '
                                   f'GEDepth-{"Adaptive" if "dynamic_pe_neck" in cfg.model else "Vanilla"}, '
                                   f'{args.height}x{args.width}, {per_gpu} img/GPU, full train step',
                       'global_batch': per_gpu * world, 'parallelism': f'dp{world}', 'last_loss': round(float(loss), 5),
                       'params': int(optimizer.arena.numel)},
        }
        prof = kernels.PROFILER.summary()
        if prof:
            # the MSDA backward is five kernels behind one entry point: rank its kernels individually (timed by HIP events
            # inside the library), so that `roofline` is about ONE kernel whose name rocprofv3 reports too
            stages = kernels.PROFILER.msda_bwd_stages()
            single = [r for r in prof if not (stages and r['name'].startswith('msda_bwd['))] + stages
            dom = max(single, key=lambda r: r['total_ms'])
            gbs = dom['bytes_per_launch'] / (dom['avg_us'] * 1e-6) / 1e9
            res['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None,
                               'avg_us': round(dom['avg_us'], 2), 'launches': dom['launches'],
                               'algorithmic_bytes_per_launch': int(dom['bytes_per_launch'])}
            res['roofline']['traffic'] = pmc_traffic(dom['name'])
            res['kernels'] = [{'name': r['name'], 'launches': r['launches'], 'avg_us': round(r['avg_us'], 2),
                               'GBps': round(r['bytes_per_launch'] / (r['avg_us'] * 1e-6) / 1e9, 1),
                               'share_of_step': round(r['total_ms'] / (1e3 * elapsed), 4)} for r in
                              sorted(prof + stages, key=lambda r: -r['total_ms'])[:24]]
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args.config, args.height, args.width)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
