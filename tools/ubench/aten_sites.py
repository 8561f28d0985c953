"""Which Python call sites launch the remaining ATen kernels of the bench step?  torch.profiler with stacks, grouped by the
innermost repo frame; run on the GPU box: python tools/ubench/aten_sites.py"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    sys.argv = ['bench.py', '--no-cpu-baseline', '--no-fp32', '--no-kernel-timing', '--graph', 'off'] + (['--config', os.environ['CFG']] if os.environ.get('CFG') else [])
    args = bench.parse()
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    use_miopen_find_db()
    use_tuned_gemms('load')
    torch.backends.cudnn.benchmark = True
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', args.config))
    cfg.model.pretrained = None
    dev = torch.device('cuda', 0)
    step, per_gpu, opt = bench.build_job(args, cfg, dev, 0, 'bf16')
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5
    from torch.autograd import DeviceType
    busy = sum(ev.device_time_total for ev in prof.events() if ev.device_type == DeviceType.CUDA)
    print(f'GPU kernel time in the profiled step: {busy / 1e3:.2f} ms; un-profiled step wall time: {wall * 1e3:.2f} ms; '
          f'kernels: {sum(1 for ev in prof.events() if ev.device_type == DeviceType.CUDA)}')
    want = ('aten::add', 'aten::add_', 'aten::copy_', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::gelu', 'aten::gelu_backward',
            'aten::sum', 'aten::mul', 'aten::div', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::clone')
    skip = ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::miopen_convolution', 'aten::convolution_backward', 'aten::mv',
            'aten::miopen_depthwise_convolution', 'aten::_convolution', 'aten::convolution', 'aten::addmv_')
    agg = defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.self_device_time_total <= 0 or ev.name in skip:
            continue
        site = 'autograd/none'
        par = ev.cpu_parent
        while par is not None and site == 'autograd/none':                 # backward ops: name the autograd node that ran them
            if par.name.startswith(('autograd::engine::evaluate_function', 'torch::autograd::')) or 'Backward' in par.name:
                site = par.name.replace('autograd::engine::evaluate_function: ', '')[:70]
            par = par.cpu_parent
        ours = [fr for fr in (ev.stack or []) if '/gedepth_amd/' in fr or 'bench.py' in fr]
        if ours:                                                                # innermost repo frame (the stack may come in either order)
            inner = ours[0] if 'bench.py' in ours[-1] else ours[-1]
            site = (inner.split('/gedepth_amd/')[-1] if '/gedepth_amd/' in inner else inner) + (' [bwd]' if site != 'autograd/none' else '')
        shp = str(ev.input_shapes)[:60] if ev.input_shapes else ''
        key = (ev.name, site, shp)
        agg[key][0] += ev.self_device_time_total
        agg[key][1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]
    for (name, site, shp), (us, n) in rows:
        print(f'{us / 1e3:7.3f} ms {n:4d}x {name:22s} {site[:70]:70s} {shp}')


if __name__ == '__main__':
    main()
