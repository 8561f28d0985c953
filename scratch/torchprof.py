"""Attribute ATen/other kernels of one training step by op + input shapes (scratch diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from gedepth_amd.depth.datasets.synthetic import synthetic_batch
from gedepth_amd.depth.models import build_depther
from gedepth_amd.mmrt.config import Config
from gedepth_amd.mmrt.optim import build_optimizer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/depthformer/depthformer_swint_v.py'))
cfg.model.pretrained = None
dev = torch.device('cuda', 0)
model = build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
model.init_weights(); model = model.to(dev).train()
opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config.get('grad_clip'))
batch = synthetic_batch(8, 352, 1120, seed=1, device=dev)
def step():
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model.train_step(batch, opt)
    out['loss'].backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=70, max_name_column_width=50, max_shapes_column_width=90))
