"""Where does the fp32 HIP path lose gradient accuracy?  HAHI neck (Swin-L widths, tiny maps like the e2e fixture) on the
GPU against the CPU oracle in float64 and float32: per-tensor l2rel of every parameter / input gradient."""
import sys
import torch
sys.path.insert(0, '/root/repo')
from oracle import gedepth_oracle as O
from oracle.fill import load_filled
from gedepth_amd.depth.models.necks.hahi import HAHIHeteroNeck

torch.set_num_threads(16)
dev = torch.device('cuda')
chans = [64, 192, 384, 768, 1536]
sizes = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
m = HAHIHeteroNeck(in_channels=chans, out_channels=chans, embedding_dim=512, scales=[1] * 5,
                   positional_encoding=dict(type='SinePositionalEncoding', num_feats=256))
sd = load_filled(m, 'hahi')
m.multi_att.dropout.p = 0.0
m.self_attn.dropout.p = 0.0
g = torch.Generator().manual_seed(7)
feats = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
G = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]


def l2rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def oracle(dtype):
    P = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.is_floating_point():
            v = v.to(dtype)
            if not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_(True)
        P['neck.' + k] = v
    xs = [f.detach().clone().to(dtype).requires_grad_(True) for f in feats]
    outs = O.hahi_neck(xs, P, train_bn=True)
    sum((o * gg.to(dtype)).sum() for o, gg in zip(outs, G)).backward()
    return outs, xs, P


o64, x64, P64 = oracle(torch.float64)
o32, x32, P32 = oracle(torch.float32)
m = m.to(dev).train()
xg = [f.detach().clone().to(dev).requires_grad_(True) for f in feats]
og = m(xg)
sum((o * gg.to(dev)).sum() for o, gg in zip(og, G)).backward()
for i in range(5):
    print(f'out{i}: HIP {l2rel(og[i], o64[i]):.2e}  cpu32 {l2rel(o32[i], o64[i]):.2e}   d_in{i}: HIP {l2rel(xg[i].grad, x64[i].grad):.2e}  '
          f'cpu32 {l2rel(x32[i].grad, x64[i].grad):.2e}')
rows = []
for k, p in m.named_parameters():
    r = P64['neck.' + k].grad
    rows.append((l2rel(p.grad, r), l2rel(P32['neck.' + k].grad, r), k))
rows.sort(reverse=True)
for a, b, k in rows[:40]:
    print(f'{k:50s} HIP {a:.2e}  cpu32 {b:.2e}')
