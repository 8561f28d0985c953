"""Writes the query geometry of the bench shapes for the standalone kernel harness (tools/ubench/msda_mm/mm_bench.cpp):
ref_cross.bin (98560 x 2 f32: sigmoid(Linear(sine)) as in the bench model), order_cross.bin (int32), ref_self.bin, order_self.bin."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
from gedepth_amd import kernels as K
torch.manual_seed(1234)
pe = SinePositionalEncoding(num_feats=256, normalize=False)     # the configs' encoding (round 4 had normalize=True here: a far more concentrated geometry than the model's)
pos = pe.grid(176, 560, 'cpu')
lin = torch.nn.Linear(512, 2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias, 0.)
ref = torch.sigmoid(lin(pos.flatten(2)[0].t())).detach()
KS = ((88, 280), (44, 140), (22, 70), (11, 35))
ref.numpy().astype(np.float32).tofile('tools/ubench/msda_mm/data/ref_cross.bin')
K.msda_ref_order(ref, KS[0]).numpy().astype(np.int32).tofile('tools/ubench/msda_mm/data/order_cross.bin')
r = []
for h, w in KS:
    gy, gx = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
    r.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
torch.cat(r, 0).numpy().astype(np.float32).tofile('tools/ubench/msda_mm/data/ref_self.bin')
K.msda_tile_order(KS, 'cpu').numpy().astype(np.int32).tofile('tools/ubench/msda_mm/data/order_self.bin')
from gedepth_amd.mmrt.bricks import msda_offset_bias
msda_offset_bias(8, 4, 8).numpy().astype(np.float32).tofile('tools/ubench/msda_mm/data/offset_bias.bin')
print('ok')
