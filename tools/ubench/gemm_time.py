"""ge_gemm_nt (csrc/gemm.hip) against the library GEMM on the token-Linear shapes of the bench workload: correctness vs float64 on the
bf16-rounded operands, time of both (HIP events, L2-cold rotation of the operands), TFLOP/s and GB/s.
  gpurun -- 'python tools/ubench/gemm_time.py [--lib path/to/variant.so] > gpurun_out/gemm_time.txt'"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from gedepth_amd import hip  # noqa: E402
from gedepth_amd.mmrt import tuning  # noqa: E402

SHAPES = [  # (M, K, N, tag)
    (788480, 512, 768, 'neck cross offsets+weights'), (788480, 512, 512, 'neck cross value/out proj'), (788480, 768, 512, 'neck cross d_query'),
    (261800, 512, 512, 'neck self proj'), (261800, 512, 768, 'neck self offsets+weights'), (261800, 768, 512, 'neck self d_query'),
    (197120, 96, 288, 's0 qkv'), (197120, 96, 384, 's0 fc1'), (197120, 384, 96, 's0 fc2'), (197120, 96, 96, 's0 proj'), (197120, 288, 96, 's0 d_qkv'),
    (49280, 192, 576, 's1 qkv'), (49280, 192, 768, 's1 fc1'), (49280, 768, 192, 's1 fc2'), (49280, 192, 192, 's1 proj'),
    (12320, 384, 1152, 's2 qkv'), (12320, 384, 1536, 's2 fc1'), (12320, 1536, 384, 's2 fc2'), (12320, 384, 384, 's2 proj'),
    (3080, 768, 2304, 's3 qkv'), (3080, 768, 3072, 's3 fc1'), (3080, 3072, 768, 's3 fc2'),
    (1000, 520, 264, 'ragged'), (257, 72, 8, 'tiny'),
]


def timeit(fn, n=10):
    for _ in range(2):
        fn(0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(n):
        fn(i)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=None)
    ap.add_argument('--epi', type=int, default=1, help='1: with bias, 0: without')
    ap.add_argument('--only', default='')
    ap.add_argument('--no-lib', action='store_true')
    args = ap.parse_args()
    lib = hip.lib() if args.lib is None else ctypes.CDLL(args.lib)
    fn = lib.ge_gemm_nt
    fn.restype, fn.argtypes = hip.SIGNATURES['ge_gemm_nt']
    dev = torch.device('cuda')
    tuning.use_tuned_gemms()            # the library side runs the committed TunableOp solutions, as the bench does
    torch.manual_seed(0)
    print(f'{"shape":34s} {"tag":28s} {"own us":>8s} {"TF/s":>7s} {"GB/s":>7s} {"lib us":>8s} {"TF/s":>7s}  ratio   max|err| (tol)')
    for M, K, N, tag in SHAPES:
        if args.only and not any(o in tag for o in args.only.split(',')):
            continue
        R = 3 if M * (K + N) * 2 < 1.5e9 else 2                 # operand rotation: three copies defeat the 256 MB Infinity Cache on the big shapes
        A = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(R)]
        W = torch.randn(N, K, device=dev).to(torch.bfloat16) * K ** -0.5
        b = torch.randn(N, device=dev)
        C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        st = torch.cuda.current_stream().cuda_stream

        def own(i):
            rc = fn(A[i % R].data_ptr(), K, W.data_ptr(), K, b.data_ptr() if args.epi else None, C[i % R].data_ptr(), N, M, N, K, hip.GE_BF16, st)
            assert rc == 0, rc
        bb = b.to(torch.bfloat16)

        def libf(i):
            torch.addmm(bb, A[i % R], W.t(), out=C[i % R]) if args.epi else torch.mm(A[i % R], W.t(), out=C[i % R])
        own(0)
        torch.cuda.synchronize()
        ref = A[0].float() @ W.float().t()                       # every element (a race in the staging shows up as a few wrong tiles)
        if args.epi:
            ref += b
        err = (C[0].float() - ref).abs().max().item()
        tol = 2 ** -8 * ref.abs().max().item() + 1e-3
        del ref
        t_own = timeit(own)
        t_lib = timeit(libf) if not args.no_lib else float('nan')
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + M * N)
        print(f'{M:7d} x {K:4d} -> {N:4d}{"":12s} {tag:28s} {t_own:8.1f} {fl / t_own / 1e6:7.1f} {by / t_own / 1e3:7.1f} {t_lib:8.1f} {fl / t_lib / 1e6:7.1f}  {t_lib / t_own:5.2f}   '
              f'{err:.3e} ({tol:.1e}) {"OK" if err <= tol else "FAIL"}', flush=True)
        del A, C


if __name__ == '__main__':
    main()
