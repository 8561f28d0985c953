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

# ROCm 7.2: hipGraph replays that overlap host -> device copies on another stream faulted ("illegal memory access") with the runtime's
# AQL-packet capture of graph kernels enabled; with it off the same runs are clean and equally fast (gedepth_amd/mmrt/graph.py).  The
# runtime reads the variable when it starts, i.e. before the first import of torch.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy peak
MFMA_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)          # 10 warm-up + 50 timed steps
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='depthformer_swint_v.py', help='file under configs/depthformer/')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: data.samples_per_gpu)')
    ap.add_argument('--height', type=int, default=352)
    ap.add_argument('--width', type=int, default=1120)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--allreduce-dtype', default='fp32', choices=['fp32', 'bf16'], help='gradient all-reduce wire format (N > 1)')
    ap.add_argument('--bucket-mb', type=float, default=None, help='DDP bucket size in MiB (default 32)')
    ap.add_argument('--attn', default='auto', choices=['auto', 'fp8'],
                    help="window attention: 'fp8' = e4m3 MFMA forward contractions (BASELINE.json configs[4]); 'auto' = bf16 MFMA")
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='on: the step (forward + losses + backward + gradient exchange + clip + AdamW) is captured in a hipGraph after three eager '
                         'steps and replayed with one launch (gedepth_amd/mmrt/graph.py); off: every kernel launched from Python; auto: on when '
                         'the step is launch-bound (<= 4 images per GPU: 1700 - 2400 launches, ~1000 of them shorter than 10 us), off when the '
                         'device is the bound anyway (8 images per GPU: 49.3 ms either way)')
    ap.add_argument('--layout', default='nhwc', choices=['nchw', 'nhwc'],
                    help='nhwc: channels-last conv stack (depth.models.utils.to_channels_last): no MIOpen layout transposes, tokens <-> maps are views')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-h2d', action='store_true', help='skip the extra pass that feeds the batch from pinned host memory every step')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip the second (HIP-event profiled) pass')
    ap.add_argument('--profile-steps', type=int, default=5, help='steps of the separate per-kernel timing pass')
    ap.add_argument('--no-fp32', action='store_true', help='skip the extra reference-precision (fp32) measurement at N=1')
    ap.add_argument('--fp32-steps', type=int, default=5)
    ap.add_argument('--gemm-tuning', default='load', choices=['off', 'load', 'tune'],
                    help='hipBLASLt/rocBLAS solution table for the Linear layers (gedepth_amd/mmrt/tuning.py)')
    ap.add_argument('--cudnn-benchmark', type=int, default=-1,
                    help='MIOpen find mode (configs: cudnn_benchmark=True): +6 %% step rate; -1 = on when the committed find-db '
                         '(gedepth_amd/tuning/miopen) is usable, else off (finding from scratch takes ~5 min on a fresh box)')
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(cfg_name, H, W):
    """the CPU oracle (oracle/gedepth_oracle.py, a port pinned to reference-generated fixtures) on the
    GPU box's host cores, fp32, batch 1: 1 warm-up + 3 timed full training steps (fwd + losses + bwd + clip 35 + AdamW)
    and 1 + 3 eval forwards.  Threads: min(host cores, 32) — the op sizes of one image do not scale further (256
    threads measured 27x slower); both numbers are reported."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from oracle import gedepth_oracle as O
    from oracle.fill import fill_state_dict
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.config import Config
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 32)
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

    def train_step():
        opt.zero_grad(set_to_none=True)
        losses, _ = O.forward_train(b['img'], b['depth_gt'], b['pe_k_gt'], P, arch, train_bn=True)
        loss = sum(v.mean() for k, v in losses.items() if 'loss' in k)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 35.0)
        opt.step()

    def eval_step():
        with torch.no_grad():
            O.encode_decode(b['img'], P, arch)

    def timed(fn, warm=1, n=3):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n
    dt_train = timed(train_step)
    dt_eval = timed(eval_step)
    return dict(value=round(1.0 / dt_train, 5), unit='img/s', cores=cores, host_cores=host_cores, cpu=_cpu_model(), kind='port',
                eval_value=round(1.0 / dt_eval, 5),
                sample=f'1 warm-up + 3 timed training steps (fwd+loss+bwd+clip+AdamW, {dt_train:.2f} s each) and 1 + 3 eval forwards '
                       f'({dt_eval:.2f} s each) on 1 image {H}x{W}, fp32, torch {torch.__version__} CPU, {cores} threads')


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


def respawn_under_launcher(args):
    """``python bench.py --gpus N`` without a launcher: re-exec under torch.distributed.run (one rank per GPU, RCCL),
    the way the reference launches training (tools/dist_train.sh:7-9)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def kernels_profiler_on():
    from gedepth_amd import kernels
    return kernels.PROFILER.on


def build_job(args, cfg, dev, rank, dtype, graph=True):
    """model + optimizer + DDP wrapper + resident synthetic batch + the step closure for one precision."""
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.mmrt.ddp import FlatDDP
    from gedepth_amd.mmrt.optim import build_optimizer
    per_gpu = args.batch or cfg.data.samples_per_gpu
    torch.manual_seed(1234)
    model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    nodrop = os.environ.get('GE_BENCH_NODROP', '')             # debugging aid: 1 = no stochastic depth and no attention dropout, path / attn = one of them off
    if nodrop:
        for m in model.modules():
            if hasattr(m, 'drop_prob') and nodrop in ('1', 'path'):
                m.drop_prob = 0.0
            if isinstance(m, torch.nn.Dropout) and nodrop in ('1', 'attn'):
                m.p = 0.0
    model = model.to(dev).train()
    if args.layout == 'nhwc':
        from gedepth_amd.depth.models.utils import to_channels_last
        to_channels_last(model)
    if args.attn == 'fp8' and dtype == 'bf16':
        for mod in model.modules():
            if hasattr(mod, 'kernel_variant'):
                mod.kernel_variant = 3
    optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
    ddp = FlatDDP(model, optimizer.arena, bucket_mb=args.bucket_mb,
                  grad_dtype=torch.bfloat16 if args.allreduce_dtype == 'bf16' else None)
    batch = synthetic_batch(per_gpu, args.height, args.width, seed=1234 + rank, device=dev)
    if 'ddad' in args.config:                                # per-sample camera heights (loading.py:923-932)
        batch['height'] = torch.full((per_gpu,), 1.56, device=dev)
    amp = dtype == 'bf16'

    def eager_step(b=None):
        optimizer.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = ddp.train_step(batch if b is None else b, optimizer)
        out['loss'].backward()
        ddp.finish()
        optimizer.step()
        return out
    step = eager_step
    if (args.graph == 'on' or (args.graph == 'auto' and per_gpu <= 4)) and graph:
        from gedepth_amd.mmrt.graph import GraphedTrainStep
        # the first 3 calls run eagerly (library kernel selection, DropPath bank, FlatDDP's arrival-order layout), the 4th captures
        gstep = GraphedTrainStep(model, optimizer, batch, amp_dtype=torch.bfloat16 if amp else None, ddp=ddp, warmup=3)

        def step(b=None):
            if kernels_profiler_on():          # the per-kernel HIP-event pass needs individual launches
                return eager_step(b)
            return gstep(b)
        step.graphed = gstep
    step.eager = eager_step
    step.batch = batch
    step.ddp = ddp
    return step, per_gpu, optimizer


def h2d_inclusive(step, steps, dev, world):
    """The same steps with the batch arriving from PINNED HOST memory every step (SURVEY.md §8d counts the H2D copy into the
    step): non-blocking copies on a side stream into two alternating device buffers, one step ahead — what tools/train.py
    does.  Reported next to `value` (which, per the bench contract, is measured with resident inputs), never instead of it."""
    host = {k: v.cpu().pin_memory() for k, v in step.batch.items() if torch.is_tensor(v)}
    rest = {k: v for k, v in step.batch.items() if not torch.is_tensor(v)}
    bufs = [{k: torch.empty_like(v, device=dev) for k, v in host.items()} for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    freed = [torch.cuda.Event(), torch.cuda.Event()]

    def stage(i):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[i])                 # the step that read this buffer two iterations ago is done
            for k, v in host.items():
                bufs[i][k].copy_(v, non_blocking=True)
            ready[i].record(copy_stream)
    for e in freed:
        e.record()
    stage(0)
    main = torch.cuda.current_stream(dev)

    def one(i):
        stage((i + 1) % 2)                                   # next batch on its way while this step runs
        main.wait_event(ready[i % 2])
        out = step(dict(bufs[i % 2], **rest))
        freed[i % 2].record(main)
        return out
    for i in range(2):
        one(i)
    fence()
    t0 = time.perf_counter()
    for i in range(2, 2 + steps):
        one(i)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    return t.item(), nbytes


T0 = time.perf_counter()


def note(msg):
    """progress on stderr (the JSON line owns stdout)"""
    print(f'[bench {time.perf_counter() - T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def fence():
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def timed_steps(step, warmup, steps, dev, world):
    for _ in range(warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item(), out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_under_launcher(args)
    # stdout carries ONE JSON line: RCCL prints its version banner there when the first communicator is created, MIOpen / hipBLASLt may
    # print too — everything written to descriptor 1 from here on goes to stderr, the result line to the saved descriptor
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    from gedepth_amd.mmrt.ddp import init_dist
    rank, local, world = init_dist('nccl')
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; there is no CPU fallback'
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    have_db = use_miopen_find_db() if args.cudnn_benchmark != 0 else False
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark == 1 or (args.cudnn_benchmark == -1 and have_db))
    use_tuned_gemms(args.gemm_tuning)

    from gedepth_amd import hip, kernels
    from gedepth_amd.mmrt.config import Config
    hip.lib()

    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', args.config))
    cfg.model.pretrained = None                             # random-init weights of the named architecture
    step, per_gpu, optimizer = build_job(args, cfg, dev, rank, args.dtype)

    note(f'model built ({args.config}, {args.dtype}); warm-up + timed region')
    kernels.FALLBACKS.clear()
    # ---- pass 1: the headline number; the per-kernel event profiler is OFF inside the timed region
    elapsed, out = timed_steps(step, args.warmup, args.steps, dev, world)
    loss = out['log_vars']['loss']
    note(f'timed region done: {1e3 * elapsed / args.steps:.2f} ms/step; H2D-inclusive pass')
    h2d_elapsed, h2d_bytes = h2d_inclusive(step, args.steps, dev, world) if not args.no_h2d else (None, 0)
    # ---- gradient exchange: one extra step with per-bucket timing (outside the timed region: the trace synchronises the device)
    ddp_info = step.ddp.describe()
    if ddp_info['active']:
        step.ddp.trace(True)
        step.eager()                                         # launched from Python: a graph replay runs no hooks, so there is nothing to stamp
        fence()
        step.ddp.trace(False)
        ddp_info['trace'] = [dict(bucket=r['bucket'], MB=round(r['bytes'] / 2 ** 20, 2), params=r['params'], launch_ms=round(r['launch_ms'], 3),
                                  done_ms=round(r['done_ms'], 3)) for r in step.ddp.bucket_trace()]
        ddp_info['note'] = ('ms since the first gradient hook of that step; launch = all-reduce enqueued on RCCL while backward continues, '
                            'done = wait() returned in finish(); an exchange that hides completes before the last bucket is launched')
    note('kernel-timing pass')
    # ---- pass 2: the same steps again with HIP events around every hand-written kernel (roofline object)
    prof, stages = [], []
    if not args.no_kernel_timing and args.profile_steps > 0:
        kernels.PROFILER.enable()
        for _ in range(args.profile_steps):
            step()
        fence()
        kernels.PROFILER.disable()
        prof = kernels.PROFILER.summary()
        stages = kernels.PROFILER.msda_bwd_stages()
    prof_ms_per_step = sum(r['total_ms'] for r in prof) / max(1, args.profile_steps)

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
                       'global_batch': per_gpu * world, 'parallelism': f'dp{world}', 'allreduce_dtype': args.allreduce_dtype, 'layout': args.layout, 'window_attention': 'fp8-e4m3 mfma fwd' if args.attn == 'fp8' else 'bf16 mfma', 'last_loss': round(float(loss), 5),
                       'params': int(optimizer.arena.numel)},
        }
        if h2d_elapsed is not None:
            res['with_h2d'] = {'value': round(total_imgs / h2d_elapsed, 3), 'unit': 'img/s', 'ms_per_step': round(1e3 * h2d_elapsed / args.steps, 3),
                               'steps': args.steps, 'bytes_per_step': int(h2d_bytes),
                               'note': 'same steps, batch copied from pinned host memory every step (non-blocking, side stream, double-buffered: '
                                       'overlaps the previous step); `value` above is with resident inputs, as the bench contract defines it'}
        res['ddp'] = ddp_info                                 # rank count as RCCL reports it, bucket layout, per-bucket overlap trace (N > 1)
        res['msda_value_choice'] = [dict(shape=list(k[:4]), mfma_kernel_levels=[l for l in range(4) if (c.mm_mask >> l) & 1],
                                         record_pipeline_levels=[l for l in range(4) if not (c.mm_mask >> l) & 1], statistics=c.last)
                                    for k, c in kernels._MM_VALUE_CHOICE.items()]      # cross-attention d_value: which kernel the run statistics chose
        res['eager_fallbacks'] = dict(kernels.FALLBACKS)      # modules that took ATen where a HIP kernel exists: must be empty
        assert not kernels.FALLBACKS, f'eager fall-backs inside the measured step: {kernels.FALLBACKS}'
        if prof:
            # the MSDA backward is several kernels behind one entry point: rank its kernels individually (timed by HIP events
            # inside the library), so that `roofline` is about ONE kernel whose name rocprofv3 reports too
            single = [r for r in prof if not (stages and r['name'].startswith(('msda_bwd[', 'msda_bwd_raw[', 'msda_bwd_value')))] + stages
            dom = max(single, key=lambda r: r['total_ms'])
            gbs = dom['bytes_per_launch'] / (dom['avg_us'] * 1e-6) / 1e9
            res['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None,
                               'avg_us': round(dom['avg_us'], 2), 'launches': dom['launches'],
                               'algorithmic_bytes_per_launch': int(dom['bytes_per_launch']),
                               'timed': f'HIP events on the launch stream, separate pass of {args.profile_steps} steps after the timed region'}
            res['roofline']['traffic'] = pmc_traffic(dom['name'].split('[')[0])
            res['roofline']['traffic_source'] = ('profiles/pmc_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                                                 'of this workload (tools/profile_bench.sh), not counters of THIS run')
            step_ms = 1e3 * elapsed / args.steps
            def krow(r):
                row = {'name': r['name'], 'launches': r['launches'], 'avg_us': round(r['avg_us'], 2),
                       'GBps': round(r['bytes_per_launch'] / (r['avg_us'] * 1e-6) / 1e9, 1),
                       'share_of_step': round(r['total_ms'] / args.profile_steps / step_ms, 4)}
                if r.get('flops_per_launch'):               # MFMA kernels (window attention): achieved TFLOP/s and fraction of the dense bf16 peak
                    tf = r['flops_per_launch'] / (r['avg_us'] * 1e-6) / 1e12
                    row['TFLOPs'] = round(tf, 2)
                    row['mfma_frac'] = round(tf / MFMA_BF16_TFLOPS, 4)
                return row
            res['kernels'] = [krow(r) for r in sorted(prof + stages, key=lambda r: -r['total_ms'])[:24]]
            res['own_kernels_ms_per_step'] = round(prof_ms_per_step, 2)
    graph_info = None
    if getattr(step, 'graphed', None) is not None:
        graph_info = dict(captured=step.graphed.graph is not None, replays=step.graphed.replays)
        step.graphed.release()
    if rank == 0:
        res['config']['hip_graph'] = graph_info or 'off'
    del step, optimizer, out
    torch.cuda.empty_cache()

    # ---- reference-precision (fp32, exact-fp32 attention off: library fp32 GEMM/conv + the same HIP kernels) line at N=1
    if world == 1 and args.dtype != 'fp32' and not args.no_fp32:
        # the committed MIOpen find-db holds the fp32 problems of the DEFAULT workload (tools/tune_tables.py --dtype fp32): find mode
        # resolves them from the db; for any other workload every fp32 convolution would be timed from scratch (minutes), so those
        # use MIOpen's immediate-mode heuristics (GE_FP32_FIND=0 forces that for the default workload too)
        default_workload = (args.config == 'depthformer_swint_v.py' and args.height == 352 and args.width == 1120 and args.batch in (None, 8)
                            and args.layout == 'nhwc')
        fp32_find = bool(torch.backends.cudnn.benchmark) and default_workload and os.environ.get('GE_FP32_FIND', '1') != '0'
        torch.backends.cudnn.benchmark = fp32_find
        note('fp32 leg')
        step32, _, opt32 = build_job(args, cfg, dev, rank, 'fp32')
        e32, o32 = timed_steps(step32, 4, args.fp32_steps, dev, world)
        if rank == 0:
            res['fp32'] = {'value': round(per_gpu * args.fp32_steps / e32, 3), 'unit': 'img/s', 'ms_per_step': round(1e3 * e32 / args.fp32_steps, 3),
                           'steps': args.fp32_steps, 'warmup': 4, 'dtype': 'fp32', 'last_loss': round(float(o32['log_vars']['loss']), 5),
                           'note': 'same workload with fp32 storage and arithmetic everywhere (the reference\'s precision; exact-fp32 window attention, MIOpen '
                                   + ('find mode from the committed find-db)' if fp32_find else 'immediate mode)')}
        if getattr(step32, 'graphed', None) is not None:
            step32.graphed.release()
        del step32, opt32, o32
        torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            note('cpu baseline (oracle, 1+3 train steps, 1+3 eval)')
            res['cpu_baseline'] = cpu_baseline(args.config, args.height, args.width)
        print(json.dumps(res), file=result_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
