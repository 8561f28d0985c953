#!/usr/bin/env python
"""bf16 accuracy / speed trade (round-4 review item 6): which parts of the bf16 step could keep fp32 storage, at what cost.

For each variant — a set of sub-modules whose forward (and therefore backward) runs OUTSIDE autocast on fp32 tensors, everything else
under bf16 autocast exactly as bench.py runs it — two numbers on the bench workload (DepthFormer-SwinT + GEDepth-Vanilla, random init):
  * training step time: 8 x 352 x 1120, full step (forward + SiLog + backward + clip + AdamW), 5 warm-up + 15 timed steps;
  * eval depth error against the all-fp32 model on the same weights at 1 x 352 x 1120: mean / max of |d - d_fp32| / max(d_fp32, 1e-3).
Variants follow the error budget of tests/test_model_gpu.py::test_bf16_eval_error_budget_per_module (neck 1.2 %, decode head 1.3 %):
  bf16                 everything under autocast (the headline mode)
  msda-core            the two deformable-attention sampling cores on fp32 value / raw projections / output (projection GEMMs stay bf16)
  head-last2           the last two UpSample stages of the DenseDepth head (88x280 -> 176x560 and the 176x560 stage) in fp32
  head                 the whole decode head in fp32
  neck                 the whole HAHI neck in fp32
  msda-core+head-last2, neck+head
    python tools/bf16_trade.py [--variants a,b] > profiles/r5_bf16_trade.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cast(o, dt):
    if torch.is_tensor(o):
        return o.to(dt) if o.is_floating_point() and o.dtype != dt else o
    if isinstance(o, (list, tuple)):
        return type(o)(cast(v, dt) for v in o)
    if isinstance(o, dict):
        return {k: cast(v, dt) for k, v in o.items()}
    return o


def fp32_island(mod, method='forward', out_dtype=None):
    """Run ``mod.<method>`` outside autocast on fp32 inputs; outputs go back in ``out_dtype`` (None: stay fp32)."""
    f = getattr(mod, method)

    def wrapped(*a, **k):
        with torch.autocast('cuda', enabled=False):
            out = f(*cast(a, torch.float32), **cast(k, torch.float32))
        return cast(out, out_dtype) if out_dtype is not None else out
    setattr(mod, method, wrapped)
    return lambda: setattr(mod, method, f)


def apply_variant(model, name):
    undo = []
    parts = name.split('+') if name != 'bf16' else []
    for p in parts:
        if p == 'msda-core':
            for att in (model.neck.multi_att, model.neck.self_attn):
                undo.append(fp32_core(att))
        elif p == 'head-last2':
            for i in (3, 4):
                undo.append(fp32_island(model.decode_head.conv_list[i]))
        elif p == 'head':
            undo.append(fp32_island(model.decode_head))
        elif p == 'neck':
            undo.append(fp32_island(model.neck))
        else:
            raise KeyError(p)
    return undo


def fp32_core(att):
    """MultiScaleDeformableAttention._attend with the SAMPLING in fp32: value_proj output, the raw offset / logit projections and the sampled
    output are fp32 tensors (fp32 gather kernels); value_proj / the query linears / output_proj remain bf16 GEMMs."""
    from gedepth_amd import kernels as K
    from gedepth_amd.mmrt import bricks
    orig = att._attend

    def attend(query, value, reference_points, spatial_shapes, key_padding_mask=None, query_shapes=None, query_order=None):
        bs, nq, _ = query.shape
        shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes)]
        v = att.value_proj(value).float().view(bs, value.shape[1], att.num_heads, -1)
        w = torch.cat((att.sampling_offsets.weight, att.attention_weights.weight), 0)
        b = torch.cat((att.sampling_offsets.bias, att.attention_weights.bias), 0)
        raw = bricks.linear_tokens(query, w, b).float()
        ref = reference_points.expand(bs, nq, att.num_levels, 2)
        with torch.autocast('cuda', enabled=False):
            out = K.ms_deform_attn_raw(v, raw, ref, shapes, query_shapes, att.num_heads, att.num_levels, att.num_points)
        return att.output_proj(out)
    att._attend = attend
    return lambda: setattr(att, '_attend', orig)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', default='bf16,msda-core,head-last2,msda-core+head-last2,head,neck,neck+head')
    ap.add_argument('--steps', type=int, default=15)
    args = ap.parse_args()
    from gedepth_amd import hip
    from gedepth_amd.depth.datasets.synthetic import synthetic_batch
    from gedepth_amd.depth.models import build_depther
    from gedepth_amd.depth.models.utils import to_channels_last
    from gedepth_amd.mmrt.config import Config
    from gedepth_amd.mmrt.optim import build_optimizer
    from gedepth_amd.mmrt.tuning import use_miopen_find_db, use_tuned_gemms
    hip.lib()
    dev = torch.device('cuda', 0)
    torch.backends.cudnn.benchmark = bool(use_miopen_find_db())
    use_tuned_gemms('load')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_v.py'))
    cfg.model.pretrained = None
    rows = []
    eval_batch = synthetic_batch(1, 352, 1120, seed=7, device=dev)
    ref = None
    for name in ['fp32'] + args.variants.split(','):
        torch.manual_seed(1234)
        model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
        model.init_weights()
        model = model.to(dev)
        to_channels_last(model)
        undo = apply_variant(model, name) if name != 'fp32' else []
        amp = name != 'fp32'
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            d = model.encode_decode(eval_batch['img'], eval_batch['img_metas']).float()
        if ref is None:
            ref = d
        rel = (d - ref).abs() / ref.abs().clamp_min(1e-3)
        model.train()
        optimizer = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
        batch = synthetic_batch(8, 352, 1120, seed=1234, device=dev)

        def step():
            optimizer.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
                out = model.train_step(batch, optimizer)
            out['loss'].backward()
            optimizer.step()
            return out
        for _ in range(5):
            out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        rows.append(dict(variant=name, ms_per_step=round(ms, 2), img_per_s=round(8e3 / ms, 1), eval_mean_rel=float(rel.mean()), eval_max_rel=float(rel.max()),
                         last_loss=round(float(out['log_vars']['loss']), 4)))
        print(f'{name:24s} {ms:8.2f} ms/step  {8e3 / ms:7.1f} img/s   eval depth vs fp32: mean {rel.mean().item():.2e} max {rel.max().item():.2e}', file=sys.stderr, flush=True)
        for u in undo:
            u()
        del model, optimizer, batch, out
        torch.cuda.empty_cache()
    base = next(r for r in rows if r['variant'] == 'bf16')
    for r in rows:
        r['slowdown_vs_bf16'] = round(r['ms_per_step'] / base['ms_per_step'] - 1, 4)
    print(json.dumps(dict(workload='DepthFormer-SwinT + GEDepth-Vanilla, random init; step 8x352x1120, eval 1x352x1120', rows=rows), indent=1))


if __name__ == '__main__':
    main()
