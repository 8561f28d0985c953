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

// MFMA phase of one staged tile for the tap group [TAP0, TAP0 + NTAP): 16 K steps (16 consecutive pixels of a tile row each), per step
// ONE dY fragment against NTAP X fragments read at the taps' pixel offsets.  Compile-time tap group: every LDS address is one base
// register + an immediate, and the fragments of step ks + 1 are read (second register set) BEFORE the MFMAs of step ks issue — with a
// single set the compiler serialised read -> s_waitcnt lgkmcnt(0) -> MFMA for each of the 80 MFMAs of a tile (round 3: 22 % MFMA-busy).
template <int TAP0, int NTAP>
__device__ __forceinline__ void wg_tile_mfma(const bf16_t* dyh, const bf16_t* xt, wg_f32x16* acc) {
  wg_bf16x8 A[2], B[2][NTAP];
#define WG_RD(P) __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, (P))), \
                                         __builtin_amdgcn_ds_read_tr16_b64_v4bf16(WG_LDS(wg_bf16x4, (P) + 4 * 32)), 0, 1, 2, 3, 4, 5, 6, 7)
#define WG_FRAGS(SET, KS)                                                                                   \
  {                                                                                                         \
    constexpr int row_ = (KS) >> 1, xh_ = ((KS) & 1) * 16;                                                  \
    A[SET] = WG_RD(dyh + (row_ * 32 + xh_) * 32);                                                           \
    _Pragma("unroll") for (int j = 0; j < NTAP; ++j) {                                                     \
      const int r_ = (TAP0 + j) / 3, s_ = (TAP0 + j) - 3 * r_;                                              \
      B[SET][j] = WG_RD(xt + ((row_ + r_) * (WG_TW + 2) + xh_ + s_) * 32);                                  \
    }                                                                                                       \
  }
#define WG_STEP(KS)                                                                                         \
  {                                                                                                         \
    if ((KS) < 15) WG_FRAGS(((KS) + 1) & 1, ((KS) + 1) & 15)                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    _Pragma("unroll") for (int j = 0; j < NTAP; ++j)                                                       \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(KS) & 1], B[(KS) & 1][j], acc[j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  }
  WG_FRAGS(0, 0)
  WG_STEP(0) WG_STEP(1) WG_STEP(2) WG_STEP(3) WG_STEP(4) WG_STEP(5) WG_STEP(6) WG_STEP(7)
  WG_STEP(8) WG_STEP(9) WG_STEP(10) WG_STEP(11) WG_STEP(12) WG_STEP(13) WG_STEP(14) WG_STEP(15)
#undef WG_STEP
#undef WG_FRAGS
#undef WG_RD
}

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
  const int total_tiles = N * tiles_y * tiles_x;                   // launcher: < 2^31
  uint4 pd[8], px[6];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  // Staging roles are the same for every tile: pixel offsets inside the tile (dY) / the halo (X), LDS slots and channel offsets are
  // computed ONCE; per tile only the tile origin (wave-uniform, scalar unit) and the image-border compares remain.  (Round 3 redid the
  // index arithmetic — two divisions by 34 and three by run-time values per piece — for every tile: 7.5 VALU instructions per MFMA by
  // SQ_INSTS_VALU / SQ_INSTS_MFMA, profiles/r3_pmc_conv3x3.txt.)
  int dy_yx[8], dy_lds[8], x_yx[6], x_lds[6];                     // (y << 16 | x) inside the tile / halo; LDS element offsets
  const int dy_ch = n0 + (tid & 7) * 8, x_ch = c0 + (tid & 3) * 8;   // part = p & 7 / p & 3 does not depend on i (256 % 8 == 0)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid + i * 256, pix = p >> 3, part = p & 7;
    dy_yx[i] = ((pix >> 5) << 16) | (pix & 31);
    dy_lds[i] = ((part >> 2) * WG_PIX + pix) * 32 + (part & 3) * 8;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = tid + i * 256, pix = p >> 2;
    const int iy = pix / (WG_TW + 2), ix = pix - iy * (WG_TW + 2);
    x_yx[i] = p < WG_X_PIECES ? ((iy << 16) | ix) : -1;
    x_lds[i] = (p >> 2) * 32 + (p & 3) * 8;
  }
  const bool dy_ch_ok = dy_ch < Cout;
#define WG_PREFETCH(T)                                                                                       \
  {                                                                                                          \
    const int t_ = __builtin_amdgcn_readfirstlane(T);                                                        \
    const int per_img = tiles_y * tiles_x;                                                                   \
    const int n_ = t_ / per_img, rem_ = t_ - n_ * per_img;                                                   \
    const int tyi = rem_ / tiles_x;                                                                          \
    const int ty0 = tyi * WG_TH, tx0 = (rem_ - tyi * tiles_x) * WG_TW;                                       \
    const bf16_t* dyn = dy + (long)n_ * H * W * Cout + dy_ch;                                                \
    const bf16_t* xn = x + (long)n_ * H * W * Cin + x_ch;                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                         \
      const int gy = ty0 + (dy_yx[i] >> 16), gx = tx0 + (dy_yx[i] & 0xffff);                                 \
      pd[i] = zero4;                                                                                         \
      if (gy < H && gx < W && dy_ch_ok) pd[i] = *(const uint4*)(dyn + (gy * W + gx) * Cout);                 \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                         \
      const int gy = ty0 + (x_yx[i] >> 16) - 1, gx = tx0 + (x_yx[i] & 0xffff) - 1;                           \
      px[i] = zero4;                                                                                         \
      if (x_yx[i] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W) px[i] = *(const uint4*)(xn + (gy * W + gx) * Cin); \
    }                                                                                                        \
  }
#define WG_PARK()                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) *(uint4*)(dy_t + dy_lds[i]) = pd[i];                        \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                           \
    if (x_yx[i] >= 0) *(uint4*)(x_t + x_lds[i]) = px[i];                                                     \
  }
  int t = blockIdx.z;
  if (t < total_tiles) {
    WG_PREFETCH(t)
    WG_PARK()
  }
  __syncthreads();
  for (; t < total_tiles; t += ksplit) {
    const bool more = t + ksplit < total_tiles;
    if (more) { WG_PREFETCH(t + ksplit) }
    // wave-uniform tap group -> two specialised copies of the phase (addresses = base + immediates)
    const bf16_t* dyh = dy_t + mb * WG_PIX * 32 + tr_pix * 32 + tr_col;
    const bf16_t* xt = x_t + tr_pix * 32 + tr_col;
    if (wv >> 1) wg_tile_mfma<5, 4>(dyh, xt, acc);
    else wg_tile_mfma<0, 5>(dyh, xt, acc);
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
  if (total >= (1L << 31) || (long)H * W * std::max(Cin, Cout) >= (1L << 31)) return GE_ERR_UNSUPPORTED;      // 32-bit tile / in-image offsets
  const int blocks_out = (Cin / 32) * ((Cout + 63) / 64);
  // K split: ONE resident round of workgroups (two per CU: 54 KB of LDS each), rounded DOWN.  Every workgroup does the same amount of
  // work, so a grid that exceeds the resident slots by a few workgroups costs a whole extra round: the round-3 rule (~1024 workgroups,
  // rounded up) launched 1026 for 576 -> 64 (18 output blocks x 57) and 1025 for 160 -> 64 (5 x 205) = three rounds where two would do.
  const int cus = ge_cu_count();                          // per device (common.h)
  if (!cus) return GE_ERR_BAD_ARG;
  long ksplit = (2L * cus) / blocks_out;
  if (ksplit > total) ksplit = total;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535 || (Cout + 63) / 64 > 65535) return GE_ERR_UNSUPPORTED;
  const dim3 grid(Cin / 32, (Cout + 63) / 64, (unsigned)ksplit);
  conv3x3_wgrad_k<<<grid, 256, 0, ge_stream(stream)>>>((const bf16_t*)x, (const bf16_t*)dy, dw, N, H, W, Cin, Cout, tiles_x, tiles_y, (int)ksplit);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
