// Swin shifted-window attention core for gfx950 — bf16 MFMA tile variant (the performance path).
//
// One wave64 per (batch, window, head).  The 49 x 32 q / k / v rows of the 7x7 window are staged in LDS
// (16-byte pieces; pad tokens take the qkv bias; rows 49..63 are zero) and every contraction runs on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//
//   forward   S^T = K Q^T   (64 keys x 64 queries as 2x2 tiles, K=32)   8 MFMA
//             O^T = V^T P^T (32 dims x 64 queries,                K=64) 8 MFMA
//   backward  S^T, dP^T = V dO^T (8+8), dQ^T = K^T dS^T (8), dV^T = dO^T P (8), dK^T = Q^T dS (8)
//
// S is computed TRANSPOSED so that, in the MFMA C/D layout (col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)),
// a lane owns one query column: soft-max statistics are lane-local plus one lane^32 exchange, and the P / dS
// accumulators are already in B-operand layout for the products contracted over keys.  The key order inside
// a K=16 step is permuted (kappa(s,hi,e) = 16 s + 4 hi + (e&3) + 8 (e>>2)) to match that register order — a
// contraction is invariant to a common permutation of A's and B's k-slots, so no cross-lane shuffles are needed.
// Products contracted over QUERIES (dV, dK) need P / dS with lane <-> key: those 64 x 32 half-tiles are transposed
// through LDS (bf16), two query tiles one after the other, reusing V's staging buffer.
//
// HBM traffic is the algorithmic minimum: q,k,v (+dO) read once, out (dqkv) written once; the op is HBM-bound
// (24.5 FLOP/B at bf16, ridge 312 FLOP/B) and MFMA only has to keep the arithmetic off the critical path.
#include <stdlib.h>
#include "common.h"
#include "window_attn.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RLD 40   // row-major bf16 tiles: 64 rows x (32 + 8 pad) -> 80-byte rows: ds_read_b128 conflict-free
#define VLD 68   // transposed V tile: 32 dims x (64 + 4 pad) keys -> 136-byte rows (8-byte aligned)

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// ---- fp8 (OCP e4m3 on gfx950) variant of the two forward contractions: v_mfma_f32_32x32x16_fp8_fp8 has the same
// (lane, k-slot) operand map as the bf16 instruction (8 values per lane and K = 16 step), so a bf16x8 fragment converts in
// registers: value * scale -> v_cvt_pk_fp8_f32.  Scales are per (window, head) and per operand: 448 / amax of the staged
// tile (the e4m3 maximum), folded back into the score scale / output multiplier.  BASELINE.json configs[4].
__device__ __forceinline__ f32x16 mfma_fp8(long a, long b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ long fp8x8_from_f32(const float v[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
__device__ __forceinline__ long fp8x8_from_bf16(bf16x8 x, float scale) {
  const uint4 u = __builtin_bit_cast(uint4, x);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16) * scale; v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u) * scale; }
  return fp8x8_from_f32(v);
}
#define FP8_MAX 448.0f
// 448 / max |x| over a staged [rows][ld] bf16 tile (first `cols` columns), wave-uniform; 1 for an all-zero tile
__device__ __forceinline__ float tile_fp8_scale(const bf16_t* tile, int ld, int rows, int cols, int lane) {
  float m = 0.f;
  for (int i = lane; i < rows * cols; i += 64) {
    const int r = i / cols, cc = i - r * cols;
    m = fmaxf(m, fabsf(bf2f(tile[r * ld + cc])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m > 0.f ? FP8_MAX / m : 1.f;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // -> v_cvt_pk_bf16_f32 (round-to-nearest-even)
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf_hw(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {   // v[base .. base+7] -> 8 bf16
  uint4 u;
  u.x = pack2(v[base + 0], v[base + 1]); u.y = pack2(v[base + 2], v[base + 3]);
  u.z = pack2(v[base + 4], v[base + 5]); u.w = pack2(v[base + 6], v[base + 7]);
  return __builtin_bit_cast(bf16x8, u);
}
// register r of the C/D layout -> row inside the 32-row tile
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// 8 bf16 of one column `col` of a row-major [.][RLD] tile, rows kappa(base, e), e = 0..7  (A operand of X^T products)
__device__ __forceinline__ bf16x8 col8(const bf16_t* tile, int base_row, int col) {
  uint32_t w[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e0 = 2 * p, e1 = 2 * p + 1;
    const int r0 = base_row + (e0 & 3) + 8 * (e0 >> 2), r1 = base_row + (e1 & 3) + 8 * (e1 >> 2);
    w[p] = (uint32_t)tile[r0 * RLD + col] | ((uint32_t)tile[r1 * RLD + col] << 16);
  }
  uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
  return __builtin_bit_cast(bf16x8, u);
}
// same but rows are CONSECUTIVE (base_row + e): used for the query-contracted products
__device__ __forceinline__ bf16x8 col8_lin(const bf16_t* tile, int base_row, int col) {
  uint32_t w[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
    w[p] = (uint32_t)tile[(base_row + 2 * p) * RLD + col] | ((uint32_t)tile[(base_row + 2 * p + 1) * RLD + col] << 16);
  uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 row8(const bf16_t* tile, int row, int col0) {   // 16-byte aligned row fragment
  return __builtin_bit_cast(bf16x8, *(const uint4*)(tile + row * RLD + col0));
}

// Stage a 49 x 32 bf16 operand row-major into LDS (rows >= 49 zero).  pad rows take `padval` (fp32 -> bf16).
__device__ __forceinline__ void stage_rm(const bf16_t* __restrict__ base, long row_stride, int col0, const int* tok,
                                         const float* __restrict__ padval, bf16_t* dst, int lane) {
  uint4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;           // 64 rows x 4 pieces of 8 bf16
    const int t = piece >> 2, part = (piece & 3) * 8;
    v[it] = make_uint4(0, 0, 0, 0);
    if (t < WT) {
      const int src = tok[t];
      if (src >= 0) {
        v[it] = *(const uint4*)(base + (long)src * row_stride + col0 + part);
      } else if (padval) {
        const float* pv = padval + col0 + part;
        v[it] = make_uint4(pack2(pv[0], pv[1]), pack2(pv[2], pv[3]), pack2(pv[4], pv[5]), pack2(pv[6], pv[7]));
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;
    const int t = piece >> 2, part = (piece & 3) * 8;
    *(uint4*)(dst + t * RLD + part) = v[it];
  }
}
// The two halves of stage_rm, for kernels that fetch the NEXT window's operands into registers while the current one is computed.
__device__ __forceinline__ void load_rm(const bf16_t* __restrict__ base, long row_stride, int col0, const int* tok,
                                        const float* __restrict__ padval, int lane, uint4 v[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;
    const int t = piece >> 2, part = (piece & 3) * 8;
    v[it] = make_uint4(0, 0, 0, 0);
    if (t < WT) {
      const int src = tok[t];
      if (src >= 0) {
        v[it] = *(const uint4*)(base + (long)src * row_stride + col0 + part);
      } else if (padval) {
        const float* pv = padval + col0 + part;
        v[it] = make_uint4(pack2(pv[0], pv[1]), pack2(pv[2], pv[3]), pack2(pv[4], pv[5]), pack2(pv[6], pv[7]));
      }
    }
  }
}
__device__ __forceinline__ void store_rm(bf16_t* dst, int lane, const uint4 v[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;
    const int t = piece >> 2, part = (piece & 3) * 8;
    *(uint4*)(dst + t * RLD + part) = v[it];
  }
}
// Stage V transposed: vt[d][key], keys >= 49 zero.
__device__ __forceinline__ void stage_vt(const bf16_t* __restrict__ base, long row_stride, int col0, const int* tok,
                                         const float* __restrict__ padval, bf16_t* vt, int lane) {
  uint4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;
    const int t = piece >> 2, part = (piece & 3) * 8;
    v[it] = make_uint4(0, 0, 0, 0);
    if (t < WT) {
      const int src = tok[t];
      if (src >= 0) {
        v[it] = *(const uint4*)(base + (long)src * row_stride + col0 + part);
      } else {
        const float* pv = padval + col0 + part;
        v[it] = make_uint4(pack2(pv[0], pv[1]), pack2(pv[2], pv[3]), pack2(pv[4], pv[5]), pack2(pv[6], pv[7]));
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = lane + it * 64;
    const int t = piece >> 2, part = (piece & 3) * 8;
    const uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) vt[(part + e) * VLD + t] = (bf16_t)((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
  }
}

struct WinMeta {            // per-window token tables in LDS
  int tok[64];
  uint32_t meta[64];        // kb | region << 16, kb = i*13 + j
};
__device__ __forceinline__ void fill_meta(const WinGeom& g, int wy, int wx, int lane, WinMeta& m) {
  int tk = -1, region = 0, kb = 0;
  if (lane < WT) {
    tk = win_token(g, wy, wx, lane, region);
    const int i = lane / WS, j = lane - i * WS;
    kb = i * 13 + j;
  }
  m.tok[lane] = tk;
  m.meta[lane] = (uint32_t)kb | ((uint32_t)region << 16);
}

// S^T tiles: acc[kt][qt] (row = key, col = query) from row-major K and Q tiles
__device__ __forceinline__ void st_tiles(const bf16_t* A_rows, const bf16_t* B_rows, int c, int hi, f32x16 acc[2][2]) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][qt][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8 a0 = row8(A_rows, c, ks * 16 + hi * 8), a1 = row8(A_rows, 32 + c, ks * 16 + hi * 8);
    const bf16x8 b0 = row8(B_rows, c, ks * 16 + hi * 8), b1 = row8(B_rows, 32 + c, ks * 16 + hi * 8);
    acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
    acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
    acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
    acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
  }
}

// fp8 version of st_tiles: operands converted fragment by fragment with the tile scales sa (A rows) / sb (B rows); the
// accumulators hold (sa * sb) * S^T
__device__ __forceinline__ void st_tiles_fp8(const bf16_t* A_rows, const bf16_t* B_rows, int c, int hi, float sa, float sb,
                                             f32x16 acc[2][2]) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][qt][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const long a0 = fp8x8_from_bf16(row8(A_rows, c, ks * 16 + hi * 8), sa), a1 = fp8x8_from_bf16(row8(A_rows, 32 + c, ks * 16 + hi * 8), sa);
    const long b0 = fp8x8_from_bf16(row8(B_rows, c, ks * 16 + hi * 8), sb), b1 = fp8x8_from_bf16(row8(B_rows, 32 + c, ks * 16 + hi * 8), sb);
    acc[0][0] = mfma_fp8(a0, b0, acc[0][0]);
    acc[0][1] = mfma_fp8(a0, b1, acc[0][1]);
    acc[1][0] = mfma_fp8(a1, b0, acc[1][0]);
    acc[1][1] = mfma_fp8(a1, b1, acc[1][1]);
  }
}

// scores -> normalised probabilities, in place.  Lane owns query columns c (qt=0) and 32+c (qt=1).
__device__ __forceinline__ void softmax_cols(f32x16 acc[2][2], const float* bias, const uint32_t* meta, int c, int hi,
                                             float scale, bool use_mask) {
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qt * 32 + c;
    const uint32_t mq = meta[q < WT ? q : 0];
    const int qb = (int)(mq & 0xffffu) + 6 * 13 + 6;      // (iq+6)*13 + jq + 6
    const int rq = (int)(mq >> 16);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + crow(r, hi);
        float s = -INFINITY;
        if (key < WT) {
          const uint32_t mk = meta[key];
          s = acc[kt][qt][r] * scale + bias[qb - (int)(mk & 0xffffu)];
          if (use_mask && (int)(mk >> 16) != rq) s += -100.0f;
        }
        acc[kt][qt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(acc[kt][qt][r] - mx);
        acc[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= inv;
  }
}

// store a (32 dims x 32 cols) C/D tile as rows of `dst`: column `col_tok` (token index) gets dims d = crow(r,hi)
__device__ __forceinline__ void store_cols(const f32x16& o, float mul, bf16_t* __restrict__ row_ptr, int hi) {
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    uint2 u;
    u.x = pack2(o[4 * gq + 0] * mul, o[4 * gq + 1] * mul);
    u.y = pack2(o[4 * gq + 2] * mul, o[4 * gq + 3] * mul);
    *(uint2*)(row_ptr + 8 * gq + 4 * hi) = u;
  }
}

// ============================================================================================ forward
struct WinSmemMfmaFwd {
  bf16_t k[64 * RLD];
  bf16_t q[64 * RLD];
  bf16_t vt[HD * VLD];
  float bias[NBIAS + 7];
  WinMeta m;
};

template <bool FP8>
__global__ void __launch_bounds__(64) window_attn_fwd_mfma_k(const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                             const float* __restrict__ bias_table, bf16_t* __restrict__ out,
                                                             WinGeom g, float scale) {
  __shared__ __attribute__((aligned(16))) WinSmemMfmaFwd sm;
  const int lane = threadIdx.x, c = lane & 31, hi = lane >> 5;
  const int item = blockIdx.x;
  const int head = item % g.nH;
  const int bw = item / g.nH;
  const int nW = g.nWh * g.nWw;
  const int b = bw / nW, win = bw - b * nW;
  const int wy = win / g.nWw, wx = win - wy * g.nWw;
  fill_meta(g, wy, wx, lane, sm.m);
  for (int i = lane; i < NBIAS; i += GE_WAVE) sm.bias[i] = bias_table[i * g.nH + head];
  __syncthreads();
  const long L = (long)g.H * g.W;
  const bf16_t* base = qkv + (long)b * L * 3 * g.C;
  stage_rm(base, 3 * g.C, head * HD, sm.m.tok, qkv_bias, sm.q, lane);
  stage_rm(base, 3 * g.C, g.C + head * HD, sm.m.tok, qkv_bias, sm.k, lane);
  stage_vt(base, 3 * g.C, 2 * g.C + head * HD, sm.m.tok, qkv_bias, sm.vt, lane);
  __syncthreads();

  f32x16 acc[2][2];
  float out_mul = 1.f;
  float v_scale = 1.f;
  if constexpr (FP8) {
    const float sk = tile_fp8_scale(sm.k, RLD, 64, HD, lane), sq = tile_fp8_scale(sm.q, RLD, 64, HD, lane);
    v_scale = tile_fp8_scale(sm.vt, VLD, HD, 64, lane);
    st_tiles_fp8(sm.k, sm.q, c, hi, sk, sq, acc);
    scale = scale / (sk * sq);
    out_mul = 1.f / (v_scale * 256.f);                          // probabilities go to fp8 as P * 256 (<= 256 < 448)
  } else {
    st_tiles(sm.k, sm.q, c, hi, acc);
  }
  const bool use_mask = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
  softmax_cols(acc, sm.bias, sm.m.meta, c, hi, scale, use_mask);

  // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; k-slot (hi,e) <-> key kappa = kt*32 + 16 s + 4 hi + (e&3) + 8 (e>>2)
  f32x16 o[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[qt][i] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kb = kt * 32 + 16 * s + 4 * hi;
      const uint2 lo = *(const uint2*)(sm.vt + c * VLD + kb), hi8 = *(const uint2*)(sm.vt + c * VLD + kb + 8);
      const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi8.x, hi8.y));
      if constexpr (FP8) {
        const long a8 = fp8x8_from_bf16(a, v_scale);
        float p0[8], p1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { p0[e] = acc[kt][0][8 * s + e] * 256.f; p1[e] = acc[kt][1][8 * s + e] * 256.f; }
        o[0] = mfma_fp8(a8, fp8x8_from_f32(p0), o[0]);
        o[1] = mfma_fp8(a8, fp8x8_from_f32(p1), o[1]);
      } else {
        o[0] = mfma_bf16(a, pack8(acc[kt][0], 8 * s), o[0]);
        o[1] = mfma_bf16(a, pack8(acc[kt][1], 8 * s), o[1]);
      }
    }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qt * 32 + c;
    if (q < WT) {
      const int dst = sm.m.tok[q];
      if (dst >= 0) store_cols(o[qt], out_mul, out + ((long)b * L + dst) * g.C + head * HD, hi);
    }
  }
}

// =========================================================================================== backward
struct WinSmemMfmaBwd {
  bf16_t k[64 * RLD];
  bf16_t q[64 * RLD];
  bf16_t v[64 * RLD];       // re-used as the [key][32 queries] transpose buffer after dP is formed
  bf16_t go[64 * RLD];
  float bias[NBIAS + 7];
  float dbias[NBIAS + 7];
  float dpad[2 * HD];
  WinMeta m;
};
#define WS_PER_WG_MFMA (NBIAS + 2 * HD)

__global__ void __launch_bounds__(64) window_attn_bwd_mfma_k(const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                             const float* __restrict__ bias_table, const bf16_t* __restrict__ gout,
                                                             bf16_t* __restrict__ dqkv, float* __restrict__ workspace,
                                                             WinGeom g, float scale, int wg_per_head) {
  __shared__ __attribute__((aligned(16))) WinSmemMfmaBwd sm;
  const int lane = threadIdx.x, c = lane & 31, hi = lane >> 5;
  const int head = blockIdx.x % g.nH;
  const int slot = blockIdx.x / g.nH;
  const int nW = g.nWh * g.nWw;
  const int n_bw = g.B * nW;
  const long L = (long)g.H * g.W;
  for (int i = lane; i < NBIAS; i += GE_WAVE) { sm.bias[i] = bias_table[i * g.nH + head]; sm.dbias[i] = 0.f; }
  sm.dpad[lane] = 0.f;
  __syncthreads();
  // bias-table gradient: dS[q][key] lands on table entry (i_q - i_k + 6) * 13 + (j_q - j_k + 6), which depends on the
  // (query, key) position inside the window only — the same for every window this persistent workgroup visits.  So the lane
  // that owns accumulator element (kt, qt, r) sums its dS over all windows in REGISTERS and scatters once at the end
  // (64 LDS float atomics per workgroup instead of per window: ds_add_f32 costs ~170 cycles per wave-instruction, which made
  // this kernel 5x slower than the forward one).
  f32x16 dB[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) dB[kt][qt][i] = 0.f;

  for (int bw = slot; bw < n_bw; bw += wg_per_head) {
    const int b = bw / nW, win = bw - b * nW;
    const int wy = win / g.nWw, wx = win - wy * g.nWw;
    fill_meta(g, wy, wx, lane, sm.m);
    __syncthreads();
    const bf16_t* base = qkv + (long)b * L * 3 * g.C;
    stage_rm(base, 3 * g.C, head * HD, sm.m.tok, qkv_bias, sm.q, lane);
    stage_rm(base, 3 * g.C, g.C + head * HD, sm.m.tok, qkv_bias, sm.k, lane);
    stage_rm(base, 3 * g.C, 2 * g.C + head * HD, sm.m.tok, qkv_bias, sm.v, lane);
    stage_rm(gout + (long)b * L * g.C, g.C, head * HD, sm.m.tok, nullptr, sm.go, lane);
    __syncthreads();

    f32x16 P[2][2], dS[2][2];
    st_tiles(sm.k, sm.q, c, hi, P);
    const bool use_mask = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
    softmax_cols(P, sm.bias, sm.m.meta, c, hi, scale, use_mask);
    st_tiles(sm.v, sm.go, c, hi, dS);                       // dP^T = V dO^T
    __syncthreads();                                        // all lanes done reading sm.v -> transpose buffer

    // dS = P (dP - delta), delta = sum_key P dP per query;  bias-table gradient
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float delta = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += P[kt][qt][r] * dS[kt][qt][r];
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ds = P[kt][qt][r] * (dS[kt][qt][r] - delta);     // exactly 0 for rows / columns >= 49 (P == 0 there)
          dS[kt][qt][r] = ds;
          dB[kt][qt][r] += ds;
        }
    }

    // dQ^T[d][query] = scale * sum_key K^T[d][key] dS^T[key][query]
    {
      f32x16 dq[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[qt][i] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 a = col8(sm.k, kt * 32 + 16 * s + 4 * hi, c);
          dq[0] = mfma_bf16(a, pack8(dS[kt][0], 8 * s), dq[0]);
          dq[1] = mfma_bf16(a, pack8(dS[kt][1], 8 * s), dq[1]);
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = qt * 32 + c;
        if (q < WT) {
          const int dst = sm.m.tok[q];
          if (dst >= 0) store_cols(dq[qt], scale, dqkv + ((long)b * L + dst) * 3 * g.C + head * HD, hi);
        }
      }
    }

    // dV^T[d][key] = sum_q dO^T[d][q] P[q][key];  dK^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dv[kt][i] = 0.f; dk[kt][i] = 0.f; }
    bf16_t* tb = sm.v;                                       // [64 keys][RLD], columns 0..31 = queries of tile qt
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {                 // pass 0: P -> dV ; pass 1: dS -> dK
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tb[(kt * 32 + crow(r, hi)) * RLD + c] = f2bf_hw(pass == 0 ? P[kt][qt][r] : dS[kt][qt][r]);
        __syncthreads();
        const bf16_t* lhs = pass == 0 ? sm.go : sm.q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 a = col8_lin(lhs, qt * 32 + ks * 16 + hi * 8, c);    // X^T[d = c][8 consecutive queries]
          const bf16x8 b0 = row8(tb, c, ks * 16 + hi * 8), b1 = row8(tb, 32 + c, ks * 16 + hi * 8);
          if (pass == 0) { dv[0] = mfma_bf16(a, b0, dv[0]); dv[1] = mfma_bf16(a, b1, dv[1]); }
          else { dk[0] = mfma_bf16(a, b0, dk[0]); dk[1] = mfma_bf16(a, b1, dk[1]); }
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = kt * 32 + c;
      if (key < WT) {
        const int dst = sm.m.tok[key];
        if (dst >= 0) {
          bf16_t* rp = dqkv + ((long)b * L + dst) * 3 * g.C + head * HD;
          store_cols(dk[kt], scale, rp + g.C, hi);
          store_cols(dv[kt], 1.f, rp + 2 * g.C, hi);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            atomicAdd(&sm.dpad[crow(r, hi)], dk[kt][r] * scale);
            atomicAdd(&sm.dpad[HD + crow(r, hi)], dv[kt][r]);
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qt * 32 + c;
    const int qb = (q / WS) * 13 + (q % WS) + 6 * 13 + 6;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + crow(r, hi);
        if (q < WT && key < WT) atomicAdd(&sm.dbias[qb - ((key / WS) * 13 + key % WS)], dB[kt][qt][r]);
      }
  }
  __syncthreads();
  float* wsp = workspace + (long)blockIdx.x * WS_PER_WG_MFMA;
  for (int i = lane; i < NBIAS; i += GE_WAVE) wsp[i] = sm.dbias[i];
  wsp[NBIAS + lane] = sm.dpad[lane];
}

// ------------------------------------------------------------------------------------------- backward, two waves per (window, head)
// The single-wave kernel above holds P, dS, dB, dQ, dK, dV for both 32-query tiles at once: 256 VGPR + 93 AGPR = one wave per SIMD, four
// (window, head) items in flight per CU, each with 25 - 30 us of exposed latency (staging by one wave, eight barriers, four LDS transposes).
// Here wave w of a 128-thread workgroup owns QUERY TILE w: scores, soft-max, dP, dS, the bias-table gradient and dQ of its 32 queries are
// wave-local (half the registers: two waves per SIMD), the four operand tiles are staged by both waves, and only the key-side sums
// dK / dV = sum over ALL queries meet: each wave sends its partial for the other wave's key tile through LDS (8 KB each way, in the space
// of the dead operand tiles) and finishes its own key tile.
__device__ __forceinline__ void st_tiles_q(const bf16_t* A_rows, const bf16_t* B_rows, int qt, int c, int hi, f32x16 acc[2]) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[kt][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8 a0 = row8(A_rows, c, ks * 16 + hi * 8), a1 = row8(A_rows, 32 + c, ks * 16 + hi * 8);
    const bf16x8 b = row8(B_rows, qt * 32 + c, ks * 16 + hi * 8);
    acc[0] = mfma_bf16(a0, b, acc[0]);
    acc[1] = mfma_bf16(a1, b, acc[1]);
  }
}
// softmax_cols for one query tile: the lane owns query column qt * 32 + c
__device__ __forceinline__ void softmax_q(f32x16 acc[2], const float* bias, const uint32_t* meta, int qt, int c, int hi, float scale,
                                          bool use_mask) {
  const int q = qt * 32 + c;
  const uint32_t mq = meta[q < WT ? q : 0];
  const int qb = (int)(mq & 0xffffu) + 6 * 13 + 6;
  const int rq = (int)(mq >> 16);
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + crow(r, hi);
      float sc = -INFINITY;
      if (key < WT) {
        const uint32_t mk = meta[key];
        sc = acc[kt][r] * scale + bias[qb - (int)(mk & 0xffffu)];
        if (use_mask && (int)(mk >> 16) != rq) sc += -100.0f;
      }
      acc[kt][r] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(acc[kt][r] - mx);
      acc[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kt][r] *= inv;
}

struct WinSmemMfmaBwd2 {
  bf16_t k[64 * RLD];       // k | q | v | go are contiguous: 20 KB, re-used for the transposes and the dK / dV exchange
  bf16_t q[64 * RLD];
  bf16_t v[64 * RLD];
  bf16_t go[64 * RLD];
  float bias[NBIAS + 7];
  float dbias[NBIAS + 7];
  float dpad[2][2 * HD];    // per wave
  WinMeta m[2];             // token tables of the current and of the next window (double buffer)
};
#ifndef WA_BWD_WAVES
#define WA_BWD_WAVES 3                      // waves per SIMD the register allocation targets: 3 = 168 VGPRs (13 spilled registers), six workgroups per CU.
                                            // Measured (tools/ubench/winattn_time.py, same session, us per launch at the four Swin stages): 2 waves + pipeline 151 / 117 / 92 / 57,
                                            // 3 waves without it 151 / 110 / 70 / 57, and with the persistent grid cut to the six resident workgroups per CU
                                            // (window_attn.hip: bwd_wg_per_head) 130 / 98 / 80 / 60 on a slower box where the 8-per-CU grid gave 153 / 120 / 79 / 61
#endif
#ifndef WA_BWD_PREFETCH
#define WA_BWD_PREFETCH 0                   // 1: the next window's operands are fetched into 32 registers under this window's arithmetic (with 3 waves per
                                            // SIMD that spills 276 registers; with 2 it gains 7 % on stage 0 — the third wave gains 25 % on stage 2)
#endif
__global__ void __launch_bounds__(128, WA_BWD_WAVES) window_attn_bwd_mfma2_k(const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                  const float* __restrict__ bias_table, const bf16_t* __restrict__ gout,
                                                                  bf16_t* __restrict__ dqkv, float* __restrict__ workspace,
                                                                  WinGeom g, float scale, int wg_per_head) {
  __shared__ __attribute__((aligned(16))) WinSmemMfmaBwd2 sm;
  const int tid = threadIdx.x, lane = tid & 63, qt = tid >> 6, c = lane & 31, hi = lane >> 5;
  const int head = blockIdx.x % g.nH;
  const int slot = blockIdx.x / g.nH;
  const int nW = g.nWh * g.nWw;
  const int n_bw = g.B * nW;
  const long L = (long)g.H * g.W;
  for (int i = tid; i < NBIAS; i += 128) { sm.bias[i] = bias_table[i * g.nH + head]; sm.dbias[i] = 0.f; }
  if (tid < 2 * HD) { sm.dpad[0][tid] = 0.f; sm.dpad[1][tid] = 0.f; }
  __syncthreads();
  f32x16 dB[2];                                              // bias-table gradient of this wave's query tile, summed over the windows
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dB[kt][i] = 0.f;

  // Software pipeline over the windows of this persistent workgroup (round 6): the operands of window i + 1 are fetched into registers
  // (2 operands x 4 pieces per lane = 32 VGPRs; wave 0: q, k; wave 1: v, dO) right after window i's tiles have been parked in LDS, and stay
  // in flight under all of window i's arithmetic — an item used to start with 2 - 3 us of exposed load latency, 7.6 us per item in total
  // at four resident workgroups per CU.  The token tables are double-buffered in LDS (one barrier less per window).
  uint4 pa[4], pb[4];
  auto fetch = [&](int bw_, const WinMeta& mm) {
    const int b_ = bw_ / nW;
    const bf16_t* base_ = qkv + (long)b_ * L * 3 * g.C;
    if (qt == 0) {
      load_rm(base_, 3 * g.C, head * HD, mm.tok, qkv_bias, lane, pa);
      load_rm(base_, 3 * g.C, g.C + head * HD, mm.tok, qkv_bias, lane, pb);
    } else {
      load_rm(base_, 3 * g.C, 2 * g.C + head * HD, mm.tok, qkv_bias, lane, pa);
      load_rm(gout + (long)b_ * L * g.C, g.C, head * HD, mm.tok, nullptr, lane, pb);
    }
  };
  auto meta_of = [&](int bw_, WinMeta& mm) {
    const int b_ = bw_ / nW, win_ = bw_ - b_ * nW;
    const int wy_ = win_ / g.nWw, wx_ = win_ - wy_ * g.nWw;
    fill_meta(g, wy_, wx_, lane, mm);
  };
  if (slot < n_bw) {
    if (qt == 0) meta_of(slot, sm.m[0]);
    __syncthreads();
    if (WA_BWD_PREFETCH) fetch(slot, sm.m[0]);
  }
  int cur = 0;
  for (int bw = slot; bw < n_bw; bw += wg_per_head, cur ^= 1) {
    const int b = bw / nW, win = bw - b * nW;
    const int wy = win / g.nWw, wx = win - wy * g.nWw;
    const WinMeta& M = sm.m[cur];
    if (!WA_BWD_PREFETCH) fetch(bw, M);                     // (A/B) no pipeline: load now, as round 5 did
    if (qt == 0) { store_rm(sm.q, lane, pa); store_rm(sm.k, lane, pb); }
    else { store_rm(sm.v, lane, pa); store_rm(sm.go, lane, pb); }
    const int bw_next = bw + wg_per_head;
    if (bw_next < n_bw && qt == 0) meta_of(bw_next, sm.m[cur ^ 1]);
    __syncthreads();
    if (WA_BWD_PREFETCH && bw_next < n_bw) fetch(bw_next, sm.m[cur ^ 1]);

    f32x16 P[2], dS[2];
    st_tiles_q(sm.k, sm.q, qt, c, hi, P);
    const bool use_mask = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
    softmax_q(P, sm.bias, M.meta, qt, c, hi, scale, use_mask);
    st_tiles_q(sm.v, sm.go, qt, c, hi, dS);                  // dP^T = V dO^T for this wave's queries
    {
      float delta = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += P[kt][r] * dS[kt][r];
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ds = P[kt][r] * (dS[kt][r] - delta);     // exactly 0 for rows / columns >= 49 (P == 0 there)
          dS[kt][r] = ds;
          dB[kt][r] += ds;
        }
    }
    // dQ^T[d][query] = scale * sum_key K^T[d][key] dS^T[key][query]
    {
      f32x16 dq;
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[i] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) dq = mfma_bf16(col8(sm.k, kt * 32 + 16 * s + 4 * hi, c), pack8(dS[kt], 8 * s), dq);
      const int q = qt * 32 + c;
      if (q < WT) {
        const int dst = M.tok[q];
        if (dst >= 0) store_cols(dq, scale, dqkv + ((long)b * L + dst) * 3 * g.C + head * HD, hi);
      }
    }
    __syncthreads();                                         // both waves are done with V (dP) and K (dQ): they become the transpose buffers

    // dV^T[d][key] = sum_q dO^T[d][q] P[q][key] and dK^T[d][key] = sum_q Q^T[d][q] dS[q][key] over ALL 64 queries, for the 32 keys of key tile
    // `qt` (round 6): each wave parks the transposed P (then dS) of ITS query tile in its own dead operand tile (wave 0: v, wave 1: k) and, after
    // the barrier, contracts BOTH waves' tiles against its own key rows — the full sums, no exchange of fp32 partials through LDS (it cost 32
    // writes + 32 reads per lane, a barrier and 32 registers)
    f32x16 dv, dk;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dv[i] = 0.f; dk[i] = 0.f; }
    bf16_t* tb = qt == 0 ? sm.v : sm.k;                      // [64 keys][RLD], columns 0..31 = the queries of this wave's tile
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {                   // pass 0: P -> dV ; pass 1: dS -> dK
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[(kt * 32 + crow(r, hi)) * RLD + c] = f2bf_hw(pass == 0 ? P[kt][r] : dS[kt][r]);
      __syncthreads();
      const bf16_t* lhs = pass == 0 ? sm.go : sm.q;
#pragma unroll
      for (int x = 0; x < 2; ++x) {                          // query tile x, parked by wave x
        const bf16_t* tbx = x == 0 ? sm.v : sm.k;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 a = col8_lin(lhs, x * 32 + ks * 16 + hi * 8, c);    // X^T[d = c][8 consecutive queries of tile x]
          const bf16x8 bb = row8(tbx, qt * 32 + c, ks * 16 + hi * 8);      // keys of this wave's key tile
          if (pass == 0) dv = mfma_bf16(a, bb, dv);
          else dk = mfma_bf16(a, bb, dk);
        }
      }
      __syncthreads();
    }
    {
      const int key = qt * 32 + c;                           // this wave finishes key tile qt
      if (key < WT) {
        const int dst = M.tok[key];
        if (dst >= 0) {
          bf16_t* rp = dqkv + ((long)b * L + dst) * 3 * g.C + head * HD;
          store_cols(dk, scale, rp + g.C, hi);
          store_cols(dv, 1.f, rp + 2 * g.C, hi);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            atomicAdd(&sm.dpad[qt][crow(r, hi)], dk[r] * scale);           // per-wave accumulators: the sum order stays fixed
            atomicAdd(&sm.dpad[qt][HD + crow(r, hi)], dv[r]);
          }
        }
      }
    }
    // (the loop-top stores of the next window's tiles must not overtake this window's last fragment reads: the second barrier of pass 1 orders them)
  }
  for (int w = 0; w < 2; ++w) {                                // wave after wave: a fixed summation order, bit-reproducible gradients
    if (qt == w) {
      const int q = qt * 32 + c;
      const int qb = (q / WS) * 13 + (q % WS) + 6 * 13 + 6;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + crow(r, hi);
          if (q < WT && key < WT) atomicAdd(&sm.dbias[qb - ((key / WS) * 13 + key % WS)], dB[kt][r]);
        }
    }
    __syncthreads();
  }
  float* wsp = workspace + (long)blockIdx.x * WS_PER_WG_MFMA;
  for (int i = tid; i < NBIAS; i += 128) wsp[i] = sm.dbias[i];
  if (tid < 2 * HD) wsp[NBIAS + tid] = sm.dpad[0][tid] + sm.dpad[1][tid];
}

extern const int ge_window_attn_mfma_available = 3;   // bit0: forward, bit1: backward

int ge_window_attn_fwd_mfma(const void* qkv, const float* qkv_bias, const float* bias_table, void* out, const WinGeom& g,
                            float scale, bool fp8, hipStream_t s) {
  const long items = (long)g.B * g.nWh * g.nWw * g.nH;
  if (fp8)
    window_attn_fwd_mfma_k<true><<<(unsigned)items, 64, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table, (bf16_t*)out, g, scale);
  else
    window_attn_fwd_mfma_k<false><<<(unsigned)items, 64, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table, (bf16_t*)out, g, scale);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
int ge_window_attn_bwd_mfma(const void* qkv, const float* qkv_bias, const float* bias_table, const void* d_out, void* d_qkv,
                            float* workspace, const WinGeom& g, float scale, int wg_per_head, hipStream_t s) {
  static const int two_wave = [] { const char* e = getenv("GE_WINATTN_BWD"); return !(e && e[0] == '1') ? 1 : 0; }();     // GE_WINATTN_BWD=1: single-wave kernel (A/B)
  if (two_wave)
    window_attn_bwd_mfma2_k<<<(unsigned)(wg_per_head * g.nH), 128, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table,
                                                                           (const bf16_t*)d_out, (bf16_t*)d_qkv, workspace, g,
                                                                           scale, wg_per_head);
  else
    window_attn_bwd_mfma_k<<<(unsigned)(wg_per_head * g.nH), 64, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table,
                                                                         (const bf16_t*)d_out, (bf16_t*)d_qkv, workspace, g,
                                                                         scale, wg_per_head);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
