#!/usr/bin/env python
"""Numbers for the LIBRARY half of the training step (MIOpen convolutions, hipBLASLt / rocBLAS / CK GEMMs): for every
convolution / matrix-multiply call of one bench step, its problem shape, device time, achieved TFLOP/s against the dense bf16
MFMA peak and achieved GB/s of algorithmic bytes (operands read once, result written once) against the HBM peak, and which of
the two rooflines binds the shape.  north_star: "each choice evidenced by rocprof HBM GB/s or MFMA utilisation".

    python tools/library_roofline.py [--config depthformer_swint_v.py] [--out profiles/r3_library_roofline.json]

Method: torch.profiler (kineto, roctracer activity records: the same kernel durations rocprofv3 --kernel-trace reports) over ONE
step after warm-up, with shapes; every GPU kernel is attributed to the innermost ATen / custom op that launched it; ops are grouped
by (op, direction, shape).  FLOPs = 2 * MACs from the shapes.  No counters are needed for this table; the MFMA-instruction
counter pass (`rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES`) is tools/profile_bench.sh's job.
"""
import argparse
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

MFMA_BF16_TFLOPS = 2500.0
HBM_GBS = 8000.0
RIDGE = MFMA_BF16_TFLOPS * 1e12 / (HBM_GBS * 1e9)          # 312 FLOP/B


def conv_cost(p, es):
    macs = p['N'] * p['Ho'] * p['Wo'] * p['K'] * (p['C'] // p['groups']) * p['R'] * p['S']
    nbytes = es * (p['N'] * p['C'] * p['H'] * p['W'] + p['K'] * (p['C'] // p['groups']) * p['R'] * p['S'] + p['N'] * p['K'] * p['Ho'] * p['Wo'])
    return 2 * macs, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='depthformer_swint_v.py')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r3_library_roofline.json'))
    ap.add_argument('--min-us', type=float, default=30.0, help='rows below this device time per step are summed into "other"')
    a = ap.parse_args()
    import bench
    sys.argv = ['bench.py', '--no-cpu-baseline', '--no-fp32', '--no-kernel-timing', '--config', a.config]
    args = bench.parse()
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    torch.backends.cudnn.benchmark = bool(use_miopen_find_db())
    use_tuned_gemms('load')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', args.config))
    cfg.model.pretrained = None
    dev = torch.device('cuda', 0)
    step, per_gpu, opt = bench.build_job(args, cfg, dev, 0, 'bf16')
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    evs = prof.events()
    total_gpu = sum(e.device_time_total for e in evs if e.device_type == DeviceType.CUDA)
    es = 2                                                               # bf16 operands
    groups = defaultdict(lambda: dict(us=0.0, calls=0, kernels=defaultdict(float)))

    def ints(v, default):
        try:
            return tuple(int(t) for t in v) if v else default
        except TypeError:
            return default
    for e in evs:
        if e.device_type == DeviceType.CUDA or e.device_time_total <= 0:
            continue
        name = e.name
        shapes = [tuple(s) if s else () for s in (e.input_shapes or [])]
        conc = list(getattr(e, 'concrete_inputs', None) or [])
        conc += [None] * 12
        key = None
        if name == 'aten::convolution' and len(shapes) >= 2 and len(shapes[0]) == 4:
            key = ('conv fwd', shapes[0], shapes[1], ints(conc[3], (1, 1)), ints(conc[4], (shapes[1][2] // 2,) * 2), (True, True))
        elif name == 'aten::convolution_backward' and len(shapes) >= 3 and len(shapes[1]) == 4:
            mask = conc[10] if isinstance(conc[10], (list, tuple)) and len(conc[10]) >= 2 else (True, True)
            key = ('conv bwd', shapes[1], shapes[2], ints(conc[4], (1, 1)), ints(conc[5], (shapes[2][2] // 2,) * 2), (bool(mask[0]), bool(mask[1])))
        elif name in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm'):
            mats = [s for s in shapes if len(s) >= 2][-2:]
            if len(mats) == 2:
                key = ('gemm ' + name[6:], mats[0], mats[1], (), (), ())
        if key is None:
            continue
        g = groups[key]
        g['us'] += e.device_time_total                       # kernels of the op and of its children (the library call + any layout copies)
        g['calls'] += 1
        for k in (e.kernels or []):
            g['kernels'][k.name[:70]] += k.duration
    rows = []
    other = 0.0
    for (kind, s0, s1, stride, pad, mask), g in groups.items():
        if kind.startswith('conv'):
            x, w = s0, s1
            p = dict(N=x[0], C=x[1], H=x[2], W=x[3], K=w[0], R=w[2], S=w[3], stride=stride[0], groups=max(1, x[1] // max(1, w[1])))
            p['Ho'] = (x[2] + 2 * pad[0] - w[2]) // stride[0] + 1
            p['Wo'] = (x[3] + 2 * pad[-1] - w[3]) // stride[-1] + 1
            f1, b1 = conv_cost(p, es)
            n_gemm = 1 if kind == 'conv fwd' else int(mask[0]) + int(mask[1])            # backward: data and / or weight gradient
            flops, nbytes = f1 * n_gemm * g['calls'], b1 * n_gemm * g['calls']
            desc = f"{w[2]}x{w[3]} s{stride[0]} {x[1]}->{w[0]} @{p['Ho']}x{p['Wo']} N{x[0]}" + ('' if kind == 'conv fwd' else f" ({'d' if mask[0] else ''}{'w' if mask[1] else ''})")
        else:
            a_, b_ = s0, s1
            batch = a_[0] if len(a_) == 3 else 1
            M, K, N = a_[-2], a_[-1], b_[-1]
            flops = 2 * batch * M * N * K * g['calls']
            nbytes = es * batch * (M * K + K * N + M * N) * g['calls']
            desc = f"{'b%d ' % batch if batch > 1 else ''}{M}x{K} @ {K}x{N}"
        if g['us'] < a.min_us:
            other += g['us']
            continue
        t = g['us'] * 1e-6
        tf = flops / t / 1e12 if flops else 0.0
        gbs = nbytes / t / 1e9 if nbytes else 0.0
        ai = flops / nbytes if nbytes else 0.0
        bound = 'mfma' if ai >= RIDGE else 'hbm'
        frac = tf / MFMA_BF16_TFLOPS if bound == 'mfma' else gbs / HBM_GBS
        top = sorted(g['kernels'].items(), key=lambda kv: -kv[1])[:2]
        rows.append(dict(op=kind, problem=desc, calls=g['calls'], us_per_step=round(g['us'], 1), GFLOP=round(flops / 1e9, 2),
                         MB=round(nbytes / 1e6, 1), flop_per_byte=round(ai, 1), TFLOPs=round(tf, 1), mfma_frac=round(tf / MFMA_BF16_TFLOPS, 4),
                         GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_GBS, 4), bound=bound, frac_of_binding_roofline=round(frac, 4),
                         kernels=[k for k, _ in top]))
    rows.sort(key=lambda r: -r['us_per_step'])
    lib_us = sum(r['us_per_step'] for r in rows) + other
    out = dict(config=a.config, workload=f'{per_gpu} x {args.height}x{args.width} bf16, one training step', gpu_kernel_ms_per_step=round(total_gpu / 1e3, 2),
               library_ms_per_step=round(lib_us / 1e3, 2), other_small_library_calls_ms=round(other / 1e3, 2),
               peaks=dict(mfma_bf16_tflops=MFMA_BF16_TFLOPS, hbm_gbs=HBM_GBS, ridge_flop_per_byte=round(RIDGE, 1)), rows=rows)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(f'GPU kernel time {total_gpu / 1e3:.2f} ms/step; library conv + GEMM {lib_us / 1e3:.2f} ms/step ({len(rows)} rows >= {a.min_us} us)')
    print(f"{'op':26s} {'problem':44s} {'calls':>5s} {'us':>8s} {'TF/s':>7s} {'%MFMA':>6s} {'GB/s':>7s} {'%HBM':>6s} {'F/B':>6s} bound")
    for r in rows[:60]:
        print(f"{r['op']:26s} {r['problem'][:44]:44s} {r['calls']:5d} {r['us_per_step']:8.1f} {r['TFLOPs']:7.1f} {100 * r['mfma_frac']:6.1f} "
              f"{r['GBps']:7.1f} {100 * r['hbm_frac']:6.1f} {r['flop_per_byte']:6.1f} {r['bound']}")


if __name__ == '__main__':
    main()
