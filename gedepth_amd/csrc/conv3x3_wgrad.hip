// Weight gradient of the 3x3 / stride 1 / pad 1 channels-last bf16 convolution on the matrix cores (gfx950).  Partner of conv3x3.hip
// (reference layers: decode_heads/densedepth_head.py:14-27, necks/hahi.py:140-165, necks/pemask_neck.py:36-42).
//
//   dW[co, r, s, ci] = sum_{n, y, x} dY[n, y, x, co] X[n, y + r - 1, x + s - 1, ci]      per tap a GEMM: M = Cout, N = Cin, K = N H W pixels
//
// The contraction runs over PIXELS, and both operands are stored pixel-major (channels-last), so an MFMA lane needs 8 consecutive
// pixels of ONE channel: exactly what gfx950's transposing LDS read delivers (ds_read_b64_tr_b16: a 16-lane group reads a
// [4 pixels][16 channels] block and lane i receives channel i of the four pixels).  With 64-byte LDS rows (32 channels) four
// consecutive pixel rows cover the 256-byte bank width once, so the reads are conflict-free for any start pixel — which is what the nine
// taps are: the same X halo tile read at nine pixel offsets, against ONE set of dY fragments.
//
// Decomposition: workgroup = (32-channel Cin chunk, 64-channel Cout block, K split); it walks its share of the 8 x 32 pixel tiles, stages
// per tile the dY tile ([2 co halves][256 px][32]) and the (8 + 2) x (32 + 2) X halo tile of its chunk ([340 px][32]), and keeps the
// 64 x 32 x 9 partial dW in registers across all of them: wave w owns Cout half (w & 1) and taps {0..4} (w < 2) or {5..8}: 5 / 4
// accumulator tiles of 32 x 32.  One fp32 atomic flush per workgroup at the end (128-byte runs along Cin); dW must be zero-filled.
#include "common.h"

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
#define WG_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

#define WG_TH 8
#define WG_TW 32
#define WG_HALO ((WG_TH + 2) * (WG_TW + 2))          // 340 halo pixels
#define WG_PIX (WG_TH * WG_TW)                        // 256 pixels = 16 K steps of 16
#define WG_DY_PIECES (WG_PIX * 8)                     // 16-byte pieces of the dY tile (64 channels per pixel)
#define WG_X_PIECES (WG_HALO * 4)

__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ dw, int N, int H,
                                                          int W, int Cin, int Cout, int tiles_x, int tiles_y, int ksplit) {
  __shared__ __attribute__((aligned(16))) bf16_t dy_t[2 * WG_PIX * 32];      // [co half][pixel][32]: 32 KB
  __shared__ __attribute__((aligned(16))) bf16_t x_t[WG_HALO * 32];          // [halo pixel][32]: 21.8 KB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
  const int mb = wv & 1, tap0 = (wv >> 1) ? 5 : 0, ntap = (wv >> 1) ? 4 : 5;
  wg_f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) acc[t] = 0.f;
  // transposing-read lane pattern: pixel row (lane >> 5) * 8 + ((lane & 15) >> 2) (+ 4 for the second read), channel piece
  const int tr_pix = (lane >> 5) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const long total_tiles = (long)N * tiles_y * tiles_x;
  uint4 pd[8], px[6];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define WG_PREFETCH(T)                                                                                       \
  {                                                                                                          \
    const long t_ = (T);                                                                                     \
    const int n_ = (int)(t_ / (tiles_y * tiles_x)), rem_ = (int)(t_ - (long)n_ * tiles_y * tiles_x);        \
    const int ty0 = (rem_ / tiles_x) * WG_TH, tx0 = (rem_ % tiles_x) * WG_TW;                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                         \
      const int p = tid + i * 256, pix = p >> 3, part = p & 7;                                               \
      const int gy = ty0 + (pix >> 5), gx = tx0 + (pix & 31);                                                \
      pd[i] = zero4;                                                                                         \
      if (gy < H && gx < W && n0 + part * 8 < Cout) pd[i] = *(const uint4*)(dy + (((long)n_ * H + gy) * W + gx) * Cout + n0 + part * 8); \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                         \
      const int p = tid + i * 256, pix = p >> 2, part = p & 3;                                               \
      const int iy = pix / (WG_TW + 2), ix = pix - iy * (WG_TW + 2);                                         \
      const int gy = ty0 + iy - 1, gx = tx0 + ix - 1;                                                        \
      px[i] = zero4;                                                                                         \
      if (p < WG_X_PIECES && gy >= 0 && gy < H && gx >= 0 && gx < W) px[i] = *(const uint4*)(x + (((long)n_ * H + gy) * W + gx) * Cin + c0 + part * 8); \
    }                                                                                                        \
  }
#define WG_PARK()                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                           \
    const int p = tid + i * 256, pix = p >> 3, part = p & 7;                                                 \
    *(uint4*)(dy_t + ((part >> 2) * WG_PIX + pix) * 32 + (part & 3) * 8) = pd[i];                            \
  }                                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                           \
    const int p = tid + i * 256;                                                                             \
    if (p < WG_X_PIECES) *(uint4*)(x_t + (p >> 2) * 32 + (p & 3) * 8) = px[i];                               \
  }
  long t = blockIdx.z;
  if (t < total_tiles) {
    WG_PREFETCH(t)
    WG_PARK()
  }
  __syncthreads();
  for (; t < total_tiles; t += ksplit) {
    const bool more = t + ksplit < total_tiles;
    if (more) { WG_PREFETCH(t + ksplit) }
    const bf16_t* dyh = dy_t + mb * WG_PIX * 32;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {                               // K step = 16 consecutive pixels of one tile row
      const int row = ks >> 1, xh = (ks & 1) * 16;
      const bf16_t* ap = dyh + (row * 32 + xh + tr_pix) * 32 + tr_col;
      const wg_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, ap));
      const wg_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, ap + 4 * 32));
      const wg_bf16x8 A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < ntap) {
          const int tap = tap0 + j, r = tap / 3, s = tap - 3 * r;
          const bf16_t* bp = x_t + ((row + r) * (WG_TW + 2) + xh + s + tr_pix) * 32 + tr_col;
          const wg_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, bp));
          const wg_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, bp + 4 * 32));
          const wg_bf16x8 Bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) { WG_PARK() }
    __syncthreads();
  }
#undef WG_PREFETCH
#undef WG_PARK
  // flush: D layout column = lane & 31 (ci), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co of the half)
  const int ci = c0 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j < ntap) {
      const int tap = tap0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = n0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < Cout) atomicAdd(dw + ((long)co * 9 + tap) * Cin + ci, acc[j][r]);
      }
    }
  }
}

// dw (Cout, 3, 3, Cin) fp32 [the (O, H, W, I) storage order of a channels-last conv weight] += sum over pixels; the caller zero-fills it.
// x (N, H, W, Cin), dy (N, H, W, Cout) bf16 channels-last.  Cin % 32 == 0, Cout % 8 == 0.
extern "C" int ge_conv3x3_nhwc_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
  if (!x || !dy || !dw || N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_BF16 || Cin % 32 || Cout % 8 || (((uintptr_t)x | (uintptr_t)dy) & 15)) return GE_ERR_UNSUPPORTED;
  if (N == 0) return GE_OK;
  const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TH - 1) / WG_TH;
  const long total = (long)N * tiles_y * tiles_x;
  const int blocks_out = (Cin / 32) * ((Cout + 63) / 64);
  // K split: ONE resident round of workgroups (two per CU: 54 KB of LDS each), rounded DOWN.  Every workgroup does the same amount of
  // work, so a grid that exceeds the resident slots by a few workgroups costs a whole extra round: the round-3 rule (~1024 workgroups,
  // rounded up) launched 1026 for 576 -> 64 (18 output blocks x 57) and 1025 for 160 -> 64 (5 x 205) = three rounds where two would do.
  static int cus = 0;
  if (!cus) { hipDeviceProp_t p; int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return GE_ERR_BAD_ARG; cus = p.multiProcessorCount; }
  long ksplit = (2L * cus) / blocks_out;
  if (ksplit > total) ksplit = total;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535 || (Cout + 63) / 64 > 65535) return GE_ERR_UNSUPPORTED;
  const dim3 grid(Cin / 32, (Cout + 63) / 64, (unsigned)ksplit);
  conv3x3_wgrad_k<<<grid, 256, 0, ge_stream(stream)>>>((const bf16_t*)x, (const bf16_t*)dy, dw, N, H, W, Cin, Cout, tiles_x, tiles_y, (int)ksplit);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
