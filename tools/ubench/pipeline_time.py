"""Host pipeline vs device pipeline cost per KITTI training sample on the toy tree."""
import os, random, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from toy_kitti import make_toy_kitti
from gedepth_amd.depth.datasets import build_dataset
from gedepth_amd.depth.datasets.gpu_pipeline import KITTIGPUPipeline, KITTIRawDataset, draw_params
from gedepth_amd.mmrt.config import Config
torch.set_num_threads(1)
root = tempfile.mkdtemp()
split = make_toy_kitti(root)
cfg = Config.fromfile('/root/repo/configs/depthformer/depthformer_a.py')
d = cfg.data.train; d.data_root, d.split = root, split
host = build_dataset(d)
raw = KITTIRawDataset(img_dir='input', ann_dir='gt_depth', split=split, data_root=root)
pipe = KITTIGPUPipeline(data_root=root, pe_source='npy')
np.random.seed(0); random.seed(0)
t0 = time.perf_counter()
for i in range(16): host[i % 4]
t_host = (time.perf_counter() - t0) / 16
t0 = time.perf_counter()
samples = [raw[i % 4] for i in range(16)]
t_decode = (time.perf_counter() - t0) / 16
for s in samples[:4]: pipe(s, draw_params())
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in samples: pipe(s, draw_params())
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 16
print(f'host pipeline (1 thread): {t_host * 1e3:.1f} ms/sample; raw decode only: {t_decode * 1e3:.1f} ms/sample; device pipeline: {t_dev * 1e3:.2f} ms/sample (incl. H2D of the uint8 image)')
