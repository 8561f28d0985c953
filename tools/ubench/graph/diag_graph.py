"""Where does a captured step diverge from the eager one?  Same model twice, same batches; per-tensor gradient / parameter differences after
each step (eager vs GraphedTrainStep with one warm-up step)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.abspath(__file__)) + '/../../..'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from gedepth_amd.depth.datasets.synthetic import synthetic_batch
from gedepth_amd.depth.models import build_depther
from gedepth_amd.mmrt.config import Config
from gedepth_amd.mmrt.graph import GraphedTrainStep
from gedepth_amd.mmrt.optim import build_optimizer
from oracle.fill import load_filled
dev = torch.device('cuda')
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'depthformer', 'depthformer_swint_a.py'))
cfg.model.pretrained = None
cfg.model.backbone.drop_path_rate = 0.0
batches = [synthetic_batch(2, 128, 160, seed=40 + i, device=dev, valid_fraction=0.3) for i in range(3)]

def make():
    torch.manual_seed(0)
    m = build_depther(cfg.model)
    m.neck.multi_att.dropout.p = 0.0; m.neck.self_attn.dropout.p = 0.0
    load_filled(m, 'graph')
    m = m.to(dev).train()
    return m, build_optimizer(m, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))

snap = {}
for mode in ('eager', 'graph'):
    model, opt = make()
    names = {id(p): n for n, p in model.named_parameters()}
    recs = []
    if mode == 'graph':
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batches[0].items()}
        gs = GraphedTrainStep(model, opt, static, amp_dtype=torch.bfloat16, warmup=1)
    for b in batches:
        if mode == 'eager':
            opt.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = model.train_step(b, opt)
            out['loss'].backward()
            opt.step()
        else:
            out = gs(b)
        torch.cuda.synchronize()
        recs.append(dict(loss=dict(out['log_vars']), grad={names[id(p)]: v.detach().clone() for p, v in zip(opt.arena.params, opt.arena.views)},
                         param={n: p.detach().clone() for n, p in model.named_parameters()}, gnorm=float(opt.last_grad_norm)))
    snap[mode] = recs
for i in range(3):
    e, g = snap['eager'][i], snap['graph'][i]
    print(f'step {i}: loss eager {e["loss"]} graph {g["loss"]}  gnorm {e["gnorm"]:.5f} / {g["gnorm"]:.5f}')
    for what in ('grad', 'param'):
        worst = sorted(((((e[what][n] - g[what][n]).norm() / (e[what][n].norm() + 1e-20)).item(), n) for n in e[what]), reverse=True)[:6]
        print(f'   worst {what} l2rel:', [(round(v, 5), n) for v, n in worst])
