// Swin shifted-window attention core for gfx950 — exact-fp32 (VALU) variant, any storage dtype.
//
// Replaces, between the qkv and proj Linears, the whole of ShiftWindowMSA.forward / WindowMSA.forward
// (depth/models/backbones/depthformer_swin.py:285-360,193-221): pad / roll / partition / mask build /
// QK^T / bias gather / softmax / AV / reverse / un-roll / crop are folded into addressing, so the
// (B*nW, nH, 49, 49) attention tensor that the reference materialises ~5x never touches HBM.
//
// One wave64 per (batch, window, head); LDS-staged 7x7 window tiles (49 x 32 q / k / v rows); lane <-> query
// for the softmax side, lane <-> key for the dK/dV side (the 49x49 tile is transposed through LDS).
// This variant does the contractions as fp32 FMA chains (bit-equivalent to an fp32-input MFMA, which runs
// at the same rate on gfx950 — MI355X_MICROARCH.md), and is the parity path; window_attn_mfma.hip holds the
// bf16 MFMA tile variant.
#include <stdlib.h>
#include "common.h"
#include "window_attn.h"

// Stage one 49x32 operand (q, k or v part of the qkv row, or d_out) into LDS as fp32, row stride `ld`.
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ base, long row_stride, int col0, const int* tok,
                                           const float* __restrict__ padval, float* dst, int ld, int lane) {
  // 49 rows x 8 pieces of 4 elements
  for (int piece = lane; piece < WT * 8; piece += GE_WAVE) {
    const int t = piece >> 3, part = (piece & 7) * 4;
    const int src = tok[t];
    float v[4];
    if (src >= 0) {
      const T* p = base + (long)src * row_stride + col0 + part;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = Io<T>::ld(p + e);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = padval ? Io<T>::rt(padval[col0 + part + e]) : 0.f;   // Linear(0) = bias, in storage precision
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[t * ld + part + e] = v[e];
  }
}

struct WinSmemFwd {
  float q[WT * QLD];
  float k[WT * HD];
  float v[WT * HD];
  float bias[NBIAS + 7];
  int tok[64];
  int reg[64];
};

// Softmax row of `lane`'s query, kept in LDS: prob_lane[key*kstride] = P[lane][key].
__device__ __forceinline__ void softmax_row(const float* qrow, const float* ks, const float* bias, const int* reg,
                                            bool use_mask, int lane, float* prob_lane, const int kstride) {
  const int iq = lane / WS, jq = lane - iq * WS;
  const int qb = (iq + 6) * 13 + jq + 6;
  const int rq = reg[lane];
  float mx = -INFINITY;
  for (int key = 0; key < WT; ++key) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += qrow[d] * ks[key * HD + d];
    const int ik = key / WS, jk = key - ik * WS;
    s += bias[qb - (ik * 13 + jk)];
    if (use_mask && reg[key] != rq) s += -100.0f;
    prob_lane[key * kstride] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int key = 0; key < WT; ++key) {
    const float e = expf(prob_lane[key * kstride] - mx);
    prob_lane[key * kstride] = e;
    sum += e;
  }
  const float inv = 1.f / sum;
  for (int key = 0; key < WT; ++key) prob_lane[key * kstride] *= inv;
}

template <typename T>
__global__ void __launch_bounds__(64) window_attn_fwd_valu_k(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                             const float* __restrict__ bias_table, T* __restrict__ out,
                                                             WinGeom g, float scale) {
  __shared__ WinSmemFwd sm;
  __shared__ float prob[WT * 64];
  const int lane = threadIdx.x;
  const int item = blockIdx.x;                       // ((b*nW + win)*nH + head)
  const int head = item % g.nH;
  const int bw = item / g.nH;
  const int nW = g.nWh * g.nWw;
  const int b = bw / nW, win = bw - b * nW;
  const int wy = win / g.nWw, wx = win - wy * g.nWw;

  if (lane < WT) { int r; sm.tok[lane] = win_token(g, wy, wx, lane, r); sm.reg[lane] = r; }
  for (int i = lane; i < NBIAS; i += GE_WAVE) sm.bias[i] = bias_table[i * g.nH + head];
  __syncthreads();
  const long L = (long)g.H * g.W;
  const T* base = qkv + (long)b * L * 3 * g.C;
  stage_rows<T>(base, 3 * g.C, head * HD, sm.tok, qkv_bias, sm.q, QLD, lane);
  stage_rows<T>(base, 3 * g.C, g.C + head * HD, sm.tok, qkv_bias, sm.k, HD, lane);
  stage_rows<T>(base, 3 * g.C, 2 * g.C + head * HD, sm.tok, qkv_bias, sm.v, HD, lane);
  __syncthreads();
  const bool use_mask = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
  if (lane < WT) {
    float qrow[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qrow[d] = sm.q[lane * QLD + d] * scale;   // q = q * self.scale (:198)
    softmax_row(qrow, sm.k, sm.bias, sm.reg, use_mask, lane, prob + lane, 64);
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int key = 0; key < WT; ++key) {
      const float p = prob[key * 64 + lane];
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += p * sm.v[key * HD + d];
    }
    const int dst = sm.tok[lane];
    if (dst >= 0) {
      T* op = out + ((long)b * L + dst) * g.C + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) Io<T>::st(op + d, o[d]);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
struct WinSmemBwd {
  float q[WT * QLD];
  float k[WT * HD];
  float v[WT * HD];
  float go[WT * QLD];
  float P[WT * PLD];     // P[q*PLD + key]
  float dS[WT * PLD];    // dS[q*PLD + key]
  float bias[NBIAS + 7];
  float dbias[NBIAS + 7];
  float dpad[2 * HD];    // pad-token k / v gradient (goes to qkv.bias)
  int tok[64];
  int reg[64];
};

// workspace layout per workgroup: [169 dbias][64 dpad]
#define WS_PER_WG (NBIAS + 2 * HD)

template <typename T>
__global__ void __launch_bounds__(64) window_attn_bwd_valu_k(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                             const float* __restrict__ bias_table, const T* __restrict__ gout,
                                                             T* __restrict__ dqkv, float* __restrict__ workspace,
                                                             WinGeom g, float scale, int wg_per_head) {
  __shared__ WinSmemBwd sm;
  const int lane = threadIdx.x;
  const int head = blockIdx.x % g.nH;          // a workgroup keeps one head: bias-grad partials stay in LDS
  const int slot = blockIdx.x / g.nH;
  const int nW = g.nWh * g.nWw;
  const int n_bw = g.B * nW;
  const long L = (long)g.H * g.W;

  for (int i = lane; i < NBIAS; i += GE_WAVE) { sm.bias[i] = bias_table[i * g.nH + head]; sm.dbias[i] = 0.f; }
  sm.dpad[lane] = 0.f;
  __syncthreads();

  for (int bw = slot; bw < n_bw; bw += wg_per_head) {
    const int b = bw / nW, win = bw - b * nW;
    const int wy = win / g.nWw, wx = win - wy * g.nWw;
    if (lane < WT) { int r; sm.tok[lane] = win_token(g, wy, wx, lane, r); sm.reg[lane] = r; }
    __syncthreads();
    const T* base = qkv + (long)b * L * 3 * g.C;
    stage_rows<T>(base, 3 * g.C, head * HD, sm.tok, qkv_bias, sm.q, QLD, lane);
    stage_rows<T>(base, 3 * g.C, g.C + head * HD, sm.tok, qkv_bias, sm.k, HD, lane);
    stage_rows<T>(base, 3 * g.C, 2 * g.C + head * HD, sm.tok, qkv_bias, sm.v, HD, lane);
    stage_rows<T>(gout + (long)b * L * g.C, g.C, head * HD, sm.tok, nullptr, sm.go, QLD, lane);
    __syncthreads();
    const bool use_mask = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
    if (lane < WT) {
      float qrow[HD], grow[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) { qrow[d] = sm.q[lane * QLD + d] * scale; grow[d] = sm.go[lane * QLD + d]; }
      softmax_row(qrow, sm.k, sm.bias, sm.reg, use_mask, lane, sm.P + lane * PLD, 1);
      // dP = dO V^T ; delta = sum_k P dP ; dS = P (dP - delta)
      float delta = 0.f;
      for (int key = 0; key < WT; ++key) {
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) dp += grow[d] * sm.v[key * HD + d];
        const float p = sm.P[lane * PLD + key];
        sm.dS[lane * PLD + key] = dp;
        delta += p * dp;
      }
      float dq[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = 0.f;
      const int iq = lane / WS, jq = lane - iq * WS;
      const int qb = (iq + 6) * 13 + jq + 6;
      for (int key = 0; key < WT; ++key) {
        const float ds = sm.P[lane * PLD + key] * (sm.dS[lane * PLD + key] - delta);
        sm.dS[lane * PLD + key] = ds;
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] += ds * sm.k[key * HD + d];
        const int ik = key / WS, jk = key - ik * WS;
        atomicAdd(&sm.dbias[qb - (ik * 13 + jk)], ds);       // distinct bins across lanes for a fixed key
      }
      const int dst = sm.tok[lane];
      if (dst >= 0) {
        T* dp = dqkv + ((long)b * L + dst) * 3 * g.C + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) Io<T>::st(dp + d, dq[d] * scale);
      }
    }
    __syncthreads();
    if (lane < WT) {   // lane <-> key
      float dk[HD], dv[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
      for (int q = 0; q < WT; ++q) {
        const float ds = sm.dS[q * PLD + lane], p = sm.P[q * PLD + lane];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
          dk[d] += ds * (sm.q[q * QLD + d] * scale);
          dv[d] += p * sm.go[q * QLD + d];
        }
      }
      const int dst = sm.tok[lane];
      if (dst >= 0) {
        T* dp = dqkv + ((long)b * L + dst) * 3 * g.C + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { Io<T>::st(dp + g.C + d, dk[d]); Io<T>::st(dp + 2 * g.C + d, dv[d]); }
      } else {
#pragma unroll
        for (int d = 0; d < HD; ++d) { atomicAdd(&sm.dpad[d], dk[d]); atomicAdd(&sm.dpad[HD + d], dv[d]); }
      }
    }
    __syncthreads();
  }
  float* wsp = workspace + (long)blockIdx.x * WS_PER_WG;
  for (int i = lane; i < NBIAS; i += GE_WAVE) wsp[i] = sm.dbias[i];
  wsp[NBIAS + lane] = sm.dpad[lane];
}

// second stage of the deterministic reduction: sum the per-workgroup partials of each head.  grid (head, 32-entry group); 8 slices of
// the workgroup range per entry, combined through LDS in a fixed order (one workgroup per head with a serial loop over up to 512
// partials took 50-120 us per call)
__global__ void __launch_bounds__(256) window_attn_bwd_reduce_k(const float* __restrict__ workspace, float* __restrict__ d_bias_table,
                                                                float* __restrict__ d_qkv_bias, int nH, int C, int wg_per_head) {
  __shared__ float part[8][32];
  const int head = blockIdx.x, j = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.y * 32 + j;
  float s = 0.f;
  if (i < WS_PER_WG) {
#pragma unroll 4
    for (int w = sl; w < wg_per_head; w += 8) s += workspace[((long)w * nH + head) * WS_PER_WG + i];
  }
  part[sl][j] = s;
  __syncthreads();
  if (sl == 0 && i < WS_PER_WG) {
#pragma unroll
    for (int q = 1; q < 8; ++q) s += part[q][j];
    if (i < NBIAS) d_bias_table[i * nH + head] = s;
    else {
      const int k = i - NBIAS;                 // 0..31 -> k part, 32..63 -> v part
      d_qkv_bias[(k < HD ? C : 2 * C) + head * HD + (k & (HD - 1))] = s;
    }
  }
  if (blockIdx.y == 0 && threadIdx.x < HD) d_qkv_bias[head * HD + threadIdx.x] = 0.f;   // pad queries receive no gradient
}

// ------------------------------------------------------------------------------------------- C ABI
static int win_geom(int B, int H, int W, int nH, int shift, WinGeom& g) {
  if (B < 0 || H <= 0 || W <= 0 || nH <= 0 || (shift != 0 && shift != WS / 2)) return GE_ERR_BAD_ARG;
  g.B = B; g.H = H; g.W = W; g.nH = nH; g.shift = shift; g.C = nH * HD;
  g.Hp = (H + WS - 1) / WS * WS; g.Wp = (W + WS - 1) / WS * WS;
  g.nWh = g.Hp / WS; g.nWw = g.Wp / WS;
  return GE_OK;
}
static int bwd_wg_per_head(const WinGeom& g) {
  long n_bw = (long)g.B * g.nWh * g.nWw;
  static const int per_cu = [] { const char* e = getenv("GE_WINATTN_WPC"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 32 ? v : 6; }();
  long want = (256L * per_cu + g.nH - 1) / g.nH;  // persistent workgroups per CU: round 6 — six are resident (three waves per SIMD, two waves each) and exactly one resident
                                                  // round is best (8: +12 - 20 % on stages 0 / 1); until round 5 four were resident and 8 measured best
  // every persistent workgroup ends with the bias-table scatter (64 LDS atomic instructions per wave) and a workspace row for the second
  // reduction stage: keep at least `min_win` windows per workgroup so that this epilogue is amortised (coarse stages have few windows)
  static const int min_win = [] { const char* e = getenv("GE_WINATTN_MINWIN"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 64 ? v : 3; }();
  const long cap = (n_bw + min_win - 1) / min_win;
  if (want > cap) want = cap;
  if (want > n_bw) want = n_bw;
  if (want < 1) want = 1;
  return (int)want;
}

// implemented in window_attn_mfma.hip
int ge_window_attn_fwd_mfma(const void* qkv, const float* qkv_bias, const float* bias_table, void* out, const WinGeom& g,
                            float scale, bool fp8, hipStream_t s);
int ge_window_attn_bwd_mfma(const void* qkv, const float* qkv_bias, const float* bias_table, const void* d_out, void* d_qkv,
                            float* workspace, const WinGeom& g, float scale, int wg_per_head, hipStream_t s);
extern const int ge_window_attn_mfma_available;

extern "C" size_t ge_window_attn_bwd_workspace(int B, int H, int W, int nH) {
  WinGeom g;
  if (win_geom(B, H, W, nH, 0, g)) return 0;
  return (size_t)bwd_wg_per_head(g) * nH * WS_PER_WG * sizeof(float);
}

extern "C" int ge_window_attn_fwd(const void* qkv, const float* qkv_bias, const float* bias_table, void* out, int B, int H,
                                  int W, int nH, int shift, float scale, int dtype, int variant, void* stream) {
  if (!qkv || !qkv_bias || !bias_table || !out) return GE_ERR_BAD_ARG;
  WinGeom g;
  int e = win_geom(B, H, W, nH, shift, g);
  if (e) return e;
  const long items = (long)B * g.nWh * g.nWw * nH;
  if (items == 0) return GE_OK;
  if (variant == 0) variant = (dtype == GE_BF16 && (ge_window_attn_mfma_available & 1)) ? 2 : 1;
  hipStream_t s = ge_stream(stream);
  if (variant == 2 || variant == 3) {                          // 3: fp8 (e4m3) MFMA contractions, BASELINE.json configs[4]
    if (dtype != GE_BF16 || !(ge_window_attn_mfma_available & 1)) return GE_ERR_UNSUPPORTED;
    return ge_window_attn_fwd_mfma(qkv, qkv_bias, bias_table, out, g, scale, variant == 3, s);
  }
  if (variant != 1) return GE_ERR_BAD_ARG;
  if (dtype == GE_F32)
    window_attn_fwd_valu_k<float><<<(unsigned)items, 64, 0, s>>>((const float*)qkv, qkv_bias, bias_table, (float*)out, g, scale);
  else if (dtype == GE_BF16)
    window_attn_fwd_valu_k<bf16_t><<<(unsigned)items, 64, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table, (bf16_t*)out, g, scale);
  else
    return GE_ERR_UNSUPPORTED;
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_window_attn_bwd(const void* qkv, const float* qkv_bias, const float* bias_table, const void* d_out,
                                  void* d_qkv, float* d_qkv_bias, float* d_bias_table, void* workspace, int B, int H, int W,
                                  int nH, int shift, float scale, int dtype, int variant, void* stream) {
  if (!qkv || !qkv_bias || !bias_table || !d_out || !d_qkv || !d_qkv_bias || !d_bias_table || !workspace) return GE_ERR_BAD_ARG;
  WinGeom g;
  int e = win_geom(B, H, W, nH, shift, g);
  if (e) return e;
  hipStream_t s = ge_stream(stream);
  if ((long)B * g.nWh * g.nWw == 0) {
    hipError_t he = hipMemsetAsync(d_bias_table, 0, sizeof(float) * NBIAS * nH, s);
    if (he == hipSuccess) he = hipMemsetAsync(d_qkv_bias, 0, sizeof(float) * 3 * g.C, s);
    return (int)he;
  }
  const int wph = bwd_wg_per_head(g);
  if (variant == 0) variant = (dtype == GE_BF16 && (ge_window_attn_mfma_available & 2)) ? 2 : 1;
  if (variant == 2 || variant == 3) {                          // the fp8 forward differentiates through the bf16 MFMA backward
    if (dtype != GE_BF16 || !(ge_window_attn_mfma_available & 2)) return GE_ERR_UNSUPPORTED;
    e = ge_window_attn_bwd_mfma(qkv, qkv_bias, bias_table, d_out, d_qkv, (float*)workspace, g, scale, wph, s);
    if (e) return e;
  } else if (variant == 1) {
    if (dtype == GE_F32)
      window_attn_bwd_valu_k<float><<<(unsigned)(wph * nH), 64, 0, s>>>((const float*)qkv, qkv_bias, bias_table, (const float*)d_out,
                                                                        (float*)d_qkv, (float*)workspace, g, scale, wph);
    else if (dtype == GE_BF16)
      window_attn_bwd_valu_k<bf16_t><<<(unsigned)(wph * nH), 64, 0, s>>>((const bf16_t*)qkv, qkv_bias, bias_table, (const bf16_t*)d_out,
                                                                         (bf16_t*)d_qkv, (float*)workspace, g, scale, wph);
    else
      return GE_ERR_UNSUPPORTED;
    GE_LAUNCH_CHECK();
  } else {
    return GE_ERR_BAD_ARG;
  }
  window_attn_bwd_reduce_k<<<dim3((unsigned)nH, (WS_PER_WG + 31) / 32), 256, 0, s>>>((const float*)workspace, d_bias_table, d_qkv_bias, nH, g.C, wph);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
