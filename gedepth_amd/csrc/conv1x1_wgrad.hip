// Weight gradient of the 1x1 convolutions on channels-last bf16 maps (gfx950): the lateral / trans_proj / conv_proj blocks of the HAHI neck
// (depth/models/necks/hahi.py:120-166: ConvModule(k = 1) on the backbone maps, 64 -> 64 ... 768 -> 512, at 176 x 560 ... 11 x 35).
//
//   dW[co, ci] = sum_rows dY[row, co] X[row, ci]          rows = N H W pixels (a channels-last map IS the token matrix)
//
// A GEMM with a huge reduction (up to 788 480 rows) and a small output (<= 768 x 768): bandwidth-bound — 64 -> 512 @176x560 reads 0.9 GB
// for 52 GFLOP — and the library kernels (CK batched bwd_weight) run it at ~2.3 TB/s (397 us).  Decomposition, wave64-first:
//   workgroup  = 8 waves, ONE per CU (grid = Ci chunks x Co chunks x K split, sized to one resident round); it owns a (<= 512 co) x (64 or
//                96 ci) block of dW in MFMA accumulators — 32 x 32 tiles dealt round-robin to the waves, <= 6 per wave — and streams its
//                share of the rows in stages of 64: with Ci <= 96 every operand byte crosses HBM exactly once
//   operands   = both are row-major with the reduction index (rows) OUTER, so an MFMA lane needs 8 consecutive rows of one channel: the
//                transposing LDS read ds_read_b64_tr_b16 on [32-channel block][row][32] images (64-byte rows), as in conv3x3_wgrad.hip
//   staging    = 16-byte global loads one stage ahead (registers), parked into LDS between two barriers; the staging roles (row, channel
//                piece, LDS slot) of a thread are computed once
//   flush      = one fp32 atomic add per accumulator element at the end (the caller zero-fills dW), 128-byte runs along ci
#include "common.h"
#include <algorithm>

typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c1_bf16x4 __attribute__((ext_vector_type(4)));
typedef float c1_f32x16 __attribute__((ext_vector_type(16)));
#define C1_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

#define C1_ROWS 64                    // rows per stage = 4 K steps of 16
#define C1_MAX_CB 16                  // 32-channel blocks of dY per workgroup (512 output channels)
#define C1_MAX_IB 3                   // 32-channel blocks of X per workgroup (96 input channels)
#define C1_TPW 6                      // accumulator tiles per wave: 16 x 3 tiles / 8 waves
#define C1_DY_PIECES 8                // 16-byte pieces per thread and stage: 64 rows x 512 channels x 2 B / 512 threads
#define C1_X_PIECES 2                 //                                      64 rows x  96 channels x 2 B / 512 threads (rounded up)

__global__ void __launch_bounds__(512, 1) conv1x1_wgrad_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ dw, long M,
                                                          int Ci, int Co, int ci_chunk, int ksplit) {
  __shared__ __attribute__((aligned(16))) bf16_t dy_t[C1_MAX_CB * C1_ROWS * 32];      // [co block][row][32]: 64 KB
  __shared__ __attribute__((aligned(16))) bf16_t x_t[C1_MAX_IB * C1_ROWS * 32];       // [ci block][row][32]: 12 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci0 = blockIdx.x * ci_chunk, co0 = blockIdx.y * (C1_MAX_CB * 32);
  const int nib = std::min(ci_chunk, Ci - ci0) >> 5, ncb = std::min(C1_MAX_CB * 32, Co - co0) >> 5;      // Ci, Co are multiples of 32
  const int T = ncb * nib;

  // staging roles: piece p of the dY stage = (row = p / (4 ncb), 8-channel part = p % (4 ncb)); of the X stage likewise with 4 nib
  int dy_row[C1_DY_PIECES], dy_col[C1_DY_PIECES], dy_lds[C1_DY_PIECES], x_row[C1_X_PIECES], x_col[C1_X_PIECES], x_lds[C1_X_PIECES];
#pragma unroll
  for (int i = 0; i < C1_DY_PIECES; ++i) {
    const int p = tid + i * 512, ppr = 4 * ncb;
    const int row = p / ppr, part = p - row * ppr;
    dy_row[i] = row < C1_ROWS ? row : -1;
    dy_col[i] = co0 + part * 8;
    dy_lds[i] = ((part >> 2) * C1_ROWS + row) * 32 + (part & 3) * 8;
  }
#pragma unroll
  for (int i = 0; i < C1_X_PIECES; ++i) {
    const int p = tid + i * 512, ppr = 4 * nib;
    const int row = p / ppr, part = p - row * ppr;
    x_row[i] = row < C1_ROWS ? row : -1;
    x_col[i] = ci0 + part * 8;
    x_lds[i] = ((part >> 2) * C1_ROWS + row) * 32 + (part & 3) * 8;
  }
  // this wave's tiles t = wv + 8 j -> (co block t / nib, ci block t % nib): LDS byte offsets of their operand images, lane part included.
  // transposing-read lane pattern (conv3x3_wgrad.hip): row (lane >> 5) * 8 + ((lane & 15) >> 2) (+ 4 for the second read), channel piece
  const int tr_off = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  int a_off[C1_TPW], b_off[C1_TPW];
#pragma unroll
  for (int j = 0; j < C1_TPW; ++j) {
    const int t = wv + 8 * j, cb = t / nib, ib = t - cb * nib;
    a_off[j] = cb * C1_ROWS * 32 + tr_off;
    b_off[j] = ib * C1_ROWS * 32 + tr_off;
  }
  c1_f32x16 acc[C1_TPW];
#pragma unroll
  for (int j = 0; j < C1_TPW; ++j) acc[j] = 0.f;

  uint4 pd[C1_DY_PIECES], px[C1_X_PIECES];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const long nstage = (M + C1_ROWS - 1) / C1_ROWS;
#define C1_PREFETCH(S)                                                                                      \
  {                                                                                                         \
    const long r0_ = (long)(S) * C1_ROWS;                                                                   \
    _Pragma("unroll") for (int i = 0; i < C1_DY_PIECES; ++i) {                                             \
      pd[i] = zero4;                                                                                        \
      if (dy_row[i] >= 0 && r0_ + dy_row[i] < M) pd[i] = *(const uint4*)(dy + (r0_ + dy_row[i]) * Co + dy_col[i]);   \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < C1_X_PIECES; ++i) {                                              \
      px[i] = zero4;                                                                                        \
      if (x_row[i] >= 0 && r0_ + x_row[i] < M) px[i] = *(const uint4*)(x + (r0_ + x_row[i]) * Ci + x_col[i]);        \
    }                                                                                                       \
  }
#define C1_PARK()                                                                                           \
  _Pragma("unroll") for (int i = 0; i < C1_DY_PIECES; ++i) { if (dy_row[i] >= 0) *(uint4*)(dy_t + dy_lds[i]) = pd[i]; }   \
  _Pragma("unroll") for (int i = 0; i < C1_X_PIECES; ++i) { if (x_row[i] >= 0) *(uint4*)(x_t + x_lds[i]) = px[i]; }
#define C1_RD(P) __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4bf16(C1_LDS(c1_bf16x4, (P))), \
                                         __builtin_amdgcn_ds_read_tr16_b64_v4bf16(C1_LDS(c1_bf16x4, (P) + 4 * 32)), 0, 1, 2, 3, 4, 5, 6, 7)
  long s = blockIdx.z;
  if (s < nstage) {
    C1_PREFETCH(s)
    C1_PARK()
  }
  __syncthreads();
  for (; s < nstage; s += ksplit) {
    const bool more = s + ksplit < nstage;
    if (more) { C1_PREFETCH(s + ksplit) }
#pragma unroll
    for (int ks = 0; ks < C1_ROWS / 16; ++ks) {
#pragma unroll
      for (int j = 0; j < C1_TPW; ++j) {
        if (wv + 8 * j < T) {                                        // wave-uniform
          const c1_bf16x8 A = C1_RD(dy_t + a_off[j] + ks * 16 * 32);
          const c1_bf16x8 B = C1_RD(x_t + b_off[j] + ks * 16 * 32);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                                 // every wave is done reading this stage
    if (more) { C1_PARK() }
    __syncthreads();
  }
#undef C1_PREFETCH
#undef C1_PARK
#undef C1_RD
  // flush: D layout column = lane & 31 (ci of the block), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co of the block)
#pragma unroll
  for (int j = 0; j < C1_TPW; ++j) {
    const int t = wv + 8 * j;
    if (t < T) {
      const int cb = t / nib, ib = t - cb * nib;
      const int ci = ci0 + ib * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        atomicAdd(dw + (long)co * Ci + ci, acc[j][r]);
      }
    }
  }
}

// dw (Cout, Cin) fp32 += dy^T x over the M = N H W rows; x (M, Cin), dy (M, Cout) bf16 row-major (channels-last maps); the caller
// zero-fills dw.  Cin % 64 == 0 or Cin % 96 == 0; Cout % 32 == 0.
extern "C" int ge_conv1x1_nhwc_wgrad(const void* x, const void* dy, float* dw, long M, int Cin, int Cout, int dtype, void* stream) {
  if (!x || !dy || !dw || M < 0 || Cin <= 0 || Cout <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_BF16 || Cout % 32 || (Cin % 64 && Cin % 96) || (((uintptr_t)x | (uintptr_t)dy) & 15)) return GE_ERR_UNSUPPORTED;
  if (M == 0) return GE_OK;
  const int ci_chunk = (Cin % 96 == 0) ? 96 : 64;
  const int n_ci = Cin / ci_chunk, n_co = (Cout + C1_MAX_CB * 32 - 1) / (C1_MAX_CB * 32);
  if (n_ci > 65535 || n_co > 65535) return GE_ERR_UNSUPPORTED;
  const int cus = ge_cu_count();                          // per device (common.h)
  if (!cus) return GE_ERR_BAD_ARG;
  const long nstage = (M + C1_ROWS - 1) / C1_ROWS;
  long ksplit = cus / ((long)n_ci * n_co);                        // ONE resident round (one workgroup per CU), rounded down
  ksplit = std::max(1L, std::min(ksplit, std::min(nstage, 65535L)));
  const dim3 grid((unsigned)n_ci, (unsigned)n_co, (unsigned)ksplit);
  conv1x1_wgrad_k<<<grid, 512, 0, ge_stream(stream)>>>((const bf16_t*)x, (const bf16_t*)dy, dw, M, Cin, Cout, ci_chunk, (int)ksplit);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
