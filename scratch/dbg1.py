import sys, os, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import gedepth_oracle as O
from gedepth_amd.kernels import ground_plane
P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01], [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
               [0.0, 0.0, 1.0, 2.745884e-03]])
Tr = np.array([[0., -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]])
H, W = 375, 1242
pe_ref, r2, num = O.ground_plane(P2, np.eye(3), Tr, H, W)
pe64, pe32 = ground_plane(r2, num, H, W, device='cuda')
a = pe64.cpu().numpy()
bad = a != pe_ref
print('mismatch', bad.sum(), 'of', bad.size)
if bad.any():
    idx = np.argwhere(bad)[:5]
    for i, j in idx:
        print(i, j, repr(a[i, j]), repr(pe_ref[i, j]), (a[i, j].view(np.int64) - pe_ref[i, j].view(np.int64)))
    u, v = np.meshgrid(range(W), range(H), indexing='xy')
    den = r2[0] * u + r2[1] * v + r2[2]
    print('r2', [repr(x) for x in r2], repr(num))
    i, j = idx[0]
    print('den', repr(den[i, j]), repr(num / den[i, j]))
