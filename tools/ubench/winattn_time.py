"""Window-attention forward / backward times at the Swin-T stage shapes of the KITTI batch (8 x 352 x 1120).
  python tools/ubench/winattn_time.py [path/to/alternative/libgedepth_hip.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gedepth_amd import hip  # noqa: E402

if len(sys.argv) > 1:
    hip.LIB_PATH = os.path.abspath(sys.argv[1])
from gedepth_amd import kernels  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    print('library:', hip.LIB_PATH)
    for (H, W, nH) in ((88, 280, 3), (44, 140, 6), (22, 70, 12), (11, 35, 24)):
        C = nH * 32
        for shift in (0, 3):
            qkv = (torch.randn(8, H * W, 3 * C, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
            qb = torch.randn(3 * C, device=dev) * 0.1
            tab = (torch.randn(169, nH, device=dev) * 0.1).requires_grad_(True)
            qb.requires_grad_(True)
            go = torch.randn(8, H * W, C, device=dev).to(torch.bfloat16)
            def run():
                out = kernels.window_attention(qkv, qb, tab, H, W, nH, shift, 32 ** -0.5)
                out.backward(go)
                qkv.grad = None
            for _ in range(3):
                run()
            kernels.PROFILER.enable()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            kernels.PROFILER.disable()
            for r in kernels.PROFILER.summary():
                if 'window_attn' in r['name']:
                    print(f"  {r['name']:50s} {r['avg_us']:8.1f} us  {r['bytes_per_launch'] / r['avg_us'] / 1e3:7.1f} GB/s")
        


if __name__ == '__main__':
    main()
