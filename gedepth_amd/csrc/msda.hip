// Multi-scale deformable attention sampling core for gfx950 (replaces mmcv's ms_deform_attn CUDA op,
// reference call sites depth/models/necks/hahi.py:279-289,316-325).
//
// Work decomposition (wave64-first): a group of 8 (bf16) / 16 (fp32) lanes owns one (batch, query, head); each lane owns
// 16 bytes of the head's 64 channels, so every bilinear corner is one fully coalesced 128 / 256-byte read of `value` and a
// wave fetches 8 / 4 rows per instruction.  A wave covers the heads of one query, whose sampling locations / attention
// weights are contiguous in memory.  The op is a gather (no contraction) -> no MFMA; forward and the d_loc / d_attw pass are
// bound by the vector-L1 gather path.  d_value is a scatter: a counting sort of (point, tile) records +
// register-tile accumulation (below), with the direct fp32-atomic scatter kept as the no-workspace fallback.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "msda.h"
#include <string.h>
#include <mutex>
#include <vector>

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[4]) {
    float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[4]) {
    uint2 t = *(const uint2*)p;
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *(uint2*)p = t;
  }
};

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// HM (head-major work order): the work index runs (image, head, query) instead of (image, query, head), so that everything
// an XCD has in flight samples ONE head of ONE image — 4.2 MB of value rows at the KITTI shape, which its 4 MB L2 mostly
// holds — instead of all 8 heads (33.5 MB).  For query sets without spatial coherence (the cross-attention at
// initialisation: reference points = sigmoid(Linear(sine embedding)), a high-frequency function of the position) the
// kernel is bound by what misses L2, not by the gather instruction rate.
__device__ __forceinline__ long msda_group(long v, int Nq, int nH, bool hm) {
  if (!hm) return v;
  const long bh = v / Nq;
  const long q = v - bh * Nq;
  const long b = bh / nH;
  return (b * Nq + q) * nH + (bh - b * nH);
}

template <typename T, bool HM>
__global__ void __launch_bounds__(256) msda_fwd_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                  const float* __restrict__ attw, T* __restrict__ out,
                                                  long n_groups, int Nv, int Nq, int nH, int L, int P) {
  constexpr int CPL = Lanes<T>::CPL, G = 64 / CPL;         // lanes per (b,q,head) group: 16 (fp32) / 8 (bf16)
  const int c0 = (threadIdx.x % G) * CPL;
  const int nh64 = nH * 64;                                 // 32-bit element offsets inside one image (launcher: Nv * nH * 64 < 2^31)
  const long grp0 = msda_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x / G) + (threadIdx.x / G);
  const long gstride = (long)gridDim.x * (blockDim.x / G);
  for (long vgrp = grp0; vgrp < n_groups; vgrp += gstride) {   // grp = (b*Nq + q)*nH + head (one trip: the grid covers all)
    const long grp = msda_group(vgrp, Nq, nH, HM);
    const int head = (int)(grp % nH);
    const long bq = grp / nH;
    const int b = (int)(bq / Nq);
    const float* lp = loc + grp * (long)(L * P * 2);
    const float* ap = attw + grp * (long)(L * P);
    const T* vb = value + ((long)b * Nv * nH + head) * 64 + c0;
    float acc[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
    for (int l = 0; l < L; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const T* vl = vb + (long)lv.start[l] * nH * 64;
      for (int p = 0; p < P; ++p) {
        const float2 xy = *(const float2*)(lp + (l * P + p) * 2);
        const float wgt = ap[l * P + p];
        const float x = xy.x * (float)Wl - 0.5f, y = xy.y * (float)Hl - 0.5f;   // grid_sample, align_corners=False
        if (!(y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl)) continue;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float ax = x - xf, ay = y - yf;
        // clamp the addresses, zero the weights of out-of-range taps: four independent loads, no branches
        const int xa = max(x0, 0), xb = min(x0 + 1, Wl - 1), ya = max(y0, 0), yb = min(y0 + 1, Hl - 1);
        const float wxa = x0 >= 0 ? 1.f - ax : 0.f, wxb = x0 + 1 < Wl ? ax : 0.f;
        const float wya = y0 >= 0 ? 1.f - ay : 0.f, wyb = y0 + 1 < Hl ? ay : 0.f;
        float v00[CPL], v01[CPL], v10[CPL], v11[CPL];
        VecL<T>::ld(vl + mul24(mul24(ya, Wl) + xa, nh64), v00);
        VecL<T>::ld(vl + mul24(mul24(ya, Wl) + xb, nh64), v01);
        VecL<T>::ld(vl + mul24(mul24(yb, Wl) + xa, nh64), v10);
        VecL<T>::ld(vl + mul24(mul24(yb, Wl) + xb, nh64), v11);
        const float w00 = wya * wxa * wgt, w01 = wya * wxb * wgt, w10 = wyb * wxa * wgt, w11 = wyb * wxb * wgt;
#pragma unroll
        for (int i = 0; i < CPL; ++i) acc[i] += w00 * v00[i] + w01 * v01[i] + w10 * v10[i] + w11 * v11[i];
      }
    }
    VecL<T>::st(out + grp * 64 + c0, acc);
  }
}

template <typename T> struct Ld1;
template <> struct Ld1<float> { static __device__ __forceinline__ float ld(const float* p) { return *p; } };
template <> struct Ld1<bf16_t> { static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); } };

__device__ __forceinline__ float readlane_f(float v, int l) {   // the builtin is typed (int, int)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Backward: one wave64 per (batch, query, head), lane == channel.  Sampling locations / weights are wave-uniform
// (fetched once, 64 + 32 floats, and broadcast with v_readlane), every tap is ONE coalesced 256-byte read of
// `value` and ONE coalesced 256-byte fp32 atomic burst into d_value (two full 128-byte lines per instruction —
// the memory-side atomic units work per line, so line-filling bursts are what bounds this kernel).
template <typename T, int P, bool VALUE_ATOMICS>
__global__ void __launch_bounds__(256) msda_bwd_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                  const float* __restrict__ attw, const T* __restrict__ gout,
                                                  float* __restrict__ d_value, float* __restrict__ d_loc,
                                                  float* __restrict__ d_attw, long n_groups, int Nv, int Nq, int nH,
                                                  int L) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * (blockDim.x >> 6);
  const int LP = L * P;
  for (long grp = wave0; grp < n_groups; grp += wstride) {      // grp = (b*Nq + q)*nH + head ; wave-uniform
    const int head = (int)(grp % nH);
    const int b = (int)((grp / nH) / Nq);
    const float* lp = loc + grp * (long)(LP * 2);
    const float* ap = attw + grp * (long)LP;
    const long vbase = ((long)b * Nv * nH + head) * 64 + lane;
    const float go = Ld1<T>::ld(gout + grp * 64 + lane);
    // per point: 4 taps (wave-uniform branch structure), partial sums over this lane's channel
#define MSDA_POINT(j_, l_, Hl_, Wl_, sv_, sx_, sy_)                                                      \
    {                                                                                                    \
      const float lx = readlane_f(locv, 2 * (j_)), ly = readlane_f(locv, 2 * (j_) + 1);                  \
      const float wgt = readlane_f(attv, (j_));                                                          \
      const float x = lx * (float)(Wl_) - 0.5f, y = ly * (float)(Hl_) - 0.5f;                            \
      float s_val = 0.f, s_dx = 0.f, s_dy = 0.f;                                                         \
      if (y > -1.f && x > -1.f && y < (float)(Hl_) && x < (float)(Wl_)) {                                \
        const float xf = floorf(x), yf = floorf(y);                                                      \
        const int x0 = (int)xf, y0 = (int)yf;                                                            \
        const float ax = x - xf, ay = y - yf, bx = 1.f - ax, by = 1.f - ay;                              \
        const long lbase = vbase + (long)lv.start[(l_)] * nH * 64;                                       \
        const float gw = go * wgt;                                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                  \
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);                                               \
          if (yy >= 0 && yy < (Hl_) && xx >= 0 && xx < (Wl_)) {                                          \
            const long o = lbase + ((long)yy * (Wl_) + xx) * nH * 64;                                    \
            const float wx = (t & 1) ? ax : bx, wy = (t >> 1) ? ay : by;                                 \
            const float gv = go * Ld1<T>::ld(value + o);                                                 \
            s_val += wy * wx * gv;                                                                       \
            s_dx += ((t & 1) ? wy : -wy) * gv;                                                           \
            s_dy += ((t >> 1) ? wx : -wx) * gv;                                                          \
            if (VALUE_ATOMICS) atomicAdd(d_value + o, gw * (wy * wx));                                   \
          }                                                                                              \
        }                                                                                                \
      }                                                                                                  \
      sv_ = s_val; sx_ = s_dx * (wgt * (float)(Wl_)); sy_ = s_dy * (wgt * (float)(Hl_));                 \
    }
    if (P == 8 && LP <= 32) {
      // 8 points per level: 24 partial sums reduced with a reduce-scatter butterfly (30 shuffles instead of 144)
      const float locv = (lane < 2 * LP) ? lp[lane] : 0.f;
      const float attv = (lane < LP) ? ap[lane] : 0.f;
      for (int l = 0; l < L; ++l) {
        const int Hl = lv.H[l], Wl = lv.W[l];
        float part[24];
#pragma unroll
        for (int p = 0; p < 8; ++p) MSDA_POINT(l * 8 + p, l, Hl, Wl, part[p], part[8 + p], part[16 + p])
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const bool up = lane & 32;
          const float send = up ? part[k] : part[k + 12], keep = up ? part[k + 12] : part[k];
          part[k] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const bool up = lane & 16;
          const float send = up ? part[k] : part[k + 6], keep = up ? part[k + 6] : part[k];
          part[k] = keep + __shfl_xor(send, 16, 64);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool up = lane & 8;
          const float send = up ? part[k] : part[k + 3], keep = up ? part[k + 3] : part[k];
          part[k] = keep + __shfl_xor(send, 8, 64);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float v = part[k];
          v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
          part[k] = v;
        }
        if ((lane & 7) == 0) {                             // lane bits 5,4,3 select which 3 of the 24 sums it holds
          const int base = ((lane >> 5) & 1) * 12 + ((lane >> 4) & 1) * 6 + ((lane >> 3) & 1) * 3;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int idx = base + k;
            const int which = idx >> 3, j = l * 8 + (idx & 7);
            if (which == 0) d_attw[grp * (long)LP + j] = part[k];
            else d_loc[(grp * (long)LP + j) * 2 + (which - 1)] = part[k];
          }
        }
      }
    } else {
      for (int c0 = 0; c0 < LP; c0 += 32) {                       // generic: 32 points per chunk, plain wave sums
        const int npt = min(32, LP - c0);
        const float locv = (lane < 2 * npt) ? lp[c0 * 2 + lane] : 0.f;
        const float attv = (lane < npt) ? ap[c0 + lane] : 0.f;
        float my_dattw = 0.f, my_dloc = 0.f;
        for (int j = 0; j < npt; ++j) {
          const int l = (c0 + j) / P;
          const int Hl = lv.H[l], Wl = lv.W[l];
          float sv, sx, sy;
          MSDA_POINT(j, l, Hl, Wl, sv, sx, sy)
          sv = wave_sum(sv); sx = wave_sum(sx); sy = wave_sum(sy);
          if (lane == j) my_dattw = sv;
          if (lane == 2 * j) my_dloc = sx;
          if (lane == 2 * j + 1) my_dloc = sy;
        }
        if (lane < npt) d_attw[grp * (long)LP + c0 + lane] = my_dattw;
        if (lane < 2 * npt) d_loc[grp * (long)(LP * 2) + c0 * 2 + lane] = my_dloc;
      }
    }
#undef MSDA_POINT
  }
}

// d_loc / d_attw only (the binned path computes d_value separately): same 16-lane-group decomposition as the forward
// kernel (4 channels per lane, one 16-byte / 8-byte load per tap and lane, four (query, head) pairs per wave), the three
// per-point sums reduced over the group with a 4-step butterfly.
// Occupancy target of the d_loc / d_attw kernel, measured A/B in one session (cross / self launch, ms): compiler default (94 VGPRs,
// 5 waves per SIMD) 6.09 / 2.00; 8 waves 8.70 / 2.84 (spills); 6: 6.25 / 2.09; **4: 5.67 / 1.87**; 3: 5.77 / 1.95; 2: 6.07 / 2.15 —
// with 128 registers the compiler keeps the row gathers of more sampling points in flight.
#ifndef MSDA_LW_WAVES
#define MSDA_LW_WAVES 4
#endif
#ifndef MSDA_LW_PRE
#define MSDA_LW_PRE 1
#endif
#define MSDA_LW_ATTR __attribute__((amdgpu_waves_per_eu(MSDA_LW_WAVES, MSDA_LW_WAVES)))
// EMIT = true (L == 4, P == 8): the kernel writes the gradient of the RAW projection outputs instead of d_loc / d_attw —
// d_off_raw = d_loc / (W_l, H_l) and d_logit_raw = attw * (d_attw - sum_{l,p} attw * d_attw) (mmcv's view / normaliser / softmax
// backward, = msda_prep_bwd_k) — in the storage type: the fp32 d_loc / d_attw tensors (2.4 GB written and read back per
// cross-attention launch) and the separate pass over them disappear.  d_loc leaves level by level; the 8 d_attw values of a level
// sit in lanes 0-2 of the group after the reduce-scatter and wait in 12 registers for the sum over all 32 points.
struct MsdaEmit { void* d_off; long off_ld; void* d_logit; long logit_ld; };
template <typename T, bool HM, bool EMIT = false>
__global__ void __launch_bounds__(256) MSDA_LW_ATTR msda_bwd_lw_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                     const float* __restrict__ attw, const T* __restrict__ gout,
                                                     float* __restrict__ d_loc, float* __restrict__ d_attw, long n_groups,
                                                     int Nv, int Nq, int nH, int L, int P, MsdaEmit em = MsdaEmit()) {
  constexpr int CPL = Lanes<T>::CPL, G = 64 / CPL;         // 16-byte loads: 16 lanes (fp32) / 8 lanes (bf16) per group
  const int sub = threadIdx.x % G;
  const int c0 = sub * CPL;
  const int nh64 = nH * 64;
  const long grp0 = msda_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x / G) + (threadIdx.x / G);
  const long gstride = (long)gridDim.x * (blockDim.x / G);
  const long iters = (n_groups + gstride - 1) / gstride;          // wave-uniform trip count: the shuffles need all lanes
  const int LP = L * P;
  for (long it = 0; it < iters; ++it) {
    const long vgrp = grp0 + it * gstride;
    const bool live = vgrp < n_groups;
    const long g_ = live ? msda_group(vgrp, Nq, nH, HM) : 0;
    const int head = (int)(g_ % nH);
    const int b = (int)((g_ / nH) / Nq);
    const float* lp = loc + g_ * (long)(LP * 2);
    const float* ap = attw + g_ * (long)LP;
    const T* vb = value + ((long)b * Nv * nH + head) * 64 + c0;
    const lw_raw_t go = *(const lw_raw_t*)(gout + g_ * 64 + c0);
    // branch-free like the forward kernel: addresses are clamped into the map and out-of-range corners / points get a zero
    // mask, so the four 16-byte loads of several points are in flight together instead of one point per round trip
#define MSDA_LW_POINT(j_, sv_, sx_, sy_)                                                                  \
    {                                                                                                     \
      const float2 xy = *(const float2*)(lp + 2 * (j_));                                                  \
      const float wgt = ap[(j_)];                                                                         \
      const float x = xy.x * (float)Wl - 0.5f, y = xy.y * (float)Hl - 0.5f;                               \
      const bool in = live && y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl;                     \
      const float xc = fminf(fmaxf(x, -1.f), (float)Wl), yc = fminf(fmaxf(y, -1.f), (float)Hl);           \
      const float xf = floorf(xc), yf = floorf(yc);                                                       \
      const int x0 = (int)xf, y0 = (int)yf;                                                               \
      const float ax = xc - xf, ay = yc - yf, bx = 1.f - ax, by = 1.f - ay;                               \
      const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);                           \
      const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);                           \
      const bool k_xa = in && x0 >= 0, k_xb = in && x0 + 1 < Wl, k_ya = y0 >= 0, k_yb = y0 + 1 < Hl;      \
      const lw_raw_t r00 = *(const lw_raw_t*)(vl + mul24(mul24(ya, Wl) + xa, nh64));                        \
      const lw_raw_t r01 = *(const lw_raw_t*)(vl + mul24(mul24(ya, Wl) + xb, nh64));                        \
      const lw_raw_t r10 = *(const lw_raw_t*)(vl + mul24(mul24(yb, Wl) + xa, nh64));                        \
      const lw_raw_t r11 = *(const lw_raw_t*)(vl + mul24(mul24(yb, Wl) + xb, nh64));                        \
      float d00 = RowDot<T>::dot(go, r00), d01 = RowDot<T>::dot(go, r01);                                 \
      float d10 = RowDot<T>::dot(go, r10), d11 = RowDot<T>::dot(go, r11);                                 \
      d00 = (k_ya && k_xa) ? d00 : 0.f; d01 = (k_ya && k_xb) ? d01 : 0.f;   /* selects: a masked corner may hold anything */ \
      d10 = (k_yb && k_xa) ? d10 : 0.f; d11 = (k_yb && k_xb) ? d11 : 0.f;                                 \
      const float s_val = by * bx * d00 + by * ax * d01 + ay * bx * d10 + ay * ax * d11;                  \
      const float s_dx = by * (d01 - d00) + ay * (d11 - d10);                                             \
      const float s_dy = bx * (d10 - d00) + ax * (d11 - d01);                                             \
      sv_ = s_val; sx_ = s_dx * (wgt * (float)Wl); sy_ = s_dy * (wgt * (float)Hl);                        \
    }
    // Owner-lane variant (P == 8, MSDA_LW_PRE): the tap arithmetic above is identical on the 8 / 16 lanes of a (query, head) group —
    // ~45 of the ~100 VALU instructions per point.  Lane `sub` (< 8) does it ONCE for point `sub` of the level (its own 8-byte
    // location: the group's 8 loads are one 64-byte segment instead of 8 broadcasts) and packs the corner-00 row index, the four
    // corner masks and the +1 column / +1 row flags into one dword; the point loop fetches {packed, frac x, frac y, weight} of point
    // p from lane p (4 lane reads) and goes straight to the row gathers.
#define MSDA_LW_OWNER(l_, pk_, oax_, oay_, owg_)                                                          \
    {                                                                                                     \
      const int j = (l_) * 8 + (sub & 7);                                                                 \
      const float2 xy = *(const float2*)(lp + 2 * j);                                                     \
      owg_ = ap[j];                                                                                       \
      const float x = xy.x * (float)Wl - 0.5f, y = xy.y * (float)Hl - 0.5f;                               \
      const bool in = live && y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl;                     \
      const float xc = fminf(fmaxf(x, -1.f), (float)Wl), yc = fminf(fmaxf(y, -1.f), (float)Hl);           \
      const float xf = floorf(xc), yf = floorf(yc);                                                       \
      const int x0 = (int)xf, y0 = (int)yf;                                                               \
      oax_ = xc - xf; oay_ = yc - yf;                                                                     \
      const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);                           \
      const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);                           \
      const bool k_xa = in && x0 >= 0, k_xb = in && x0 + 1 < Wl, k_ya = y0 >= 0, k_yb = y0 + 1 < Hl;      \
      const int m = ((k_ya && k_xa) ? 1 : 0) | ((k_ya && k_xb) ? 2 : 0) | ((k_yb && k_xa) ? 4 : 0) | ((k_yb && k_xb) ? 8 : 0); \
      pk_ = (mul24(ya, Wl) + xa) | (m << 26) | ((xb - xa) << 30) | ((yb - ya) << 31);                     \
    }
#define MSDA_LW_POINT_PRE(p_, pk_, oax_, oay_, owg_, sv_, sx_, sy_)                                       \
    {                                                                                                     \
      const int k = __shfl(pk_, (p_), G);                                                                 \
      const float ax = __shfl(oax_, (p_), G), ay = __shfl(oay_, (p_), G), wgt = __shfl(owg_, (p_), G);    \
      const float bx = 1.f - ax, by = 1.f - ay;                                                           \
      const int i00 = k & 0x03ffffff, dx = (k >> 30) & 1;                                                 \
      const int i10 = i00 + ((k >> 31) & Wl);                                                             \
      const lw_raw_t r00 = *(const lw_raw_t*)(vl + mul24(i00, nh64));                                     \
      const lw_raw_t r01 = *(const lw_raw_t*)(vl + mul24(i00 + dx, nh64));                                \
      const lw_raw_t r10 = *(const lw_raw_t*)(vl + mul24(i10, nh64));                                     \
      const lw_raw_t r11 = *(const lw_raw_t*)(vl + mul24(i10 + dx, nh64));                                \
      float d00 = RowDot<T>::dot(go, r00), d01 = RowDot<T>::dot(go, r01);                                 \
      float d10 = RowDot<T>::dot(go, r10), d11 = RowDot<T>::dot(go, r11);                                 \
      d00 = (k & (1 << 26)) ? d00 : 0.f; d01 = (k & (2 << 26)) ? d01 : 0.f;                               \
      d10 = (k & (4 << 26)) ? d10 : 0.f; d11 = (k & (8 << 26)) ? d11 : 0.f;                               \
      const float s_val = by * bx * d00 + by * ax * d01 + ay * bx * d10 + ay * ax * d11;                  \
      const float s_dx = by * (d01 - d00) + ay * (d11 - d10);                                             \
      const float s_dy = bx * (d10 - d00) + ax * (d11 - d01);                                             \
      sv_ = s_val; sx_ = s_dx * (wgt * (float)Wl); sy_ = s_dy * (wgt * (float)Hl);                        \
    }
    float ka[4];                                      // EMIT: d_attw of point `sub` at each level (static indices: the branches below are uniform)
    for (int l = 0; l < L; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const T* vl = vb + (long)lv.start[l] * nH * 64;
      if (P == 8) {
        // 24 partial sums per level, reduce-scattered over the lane group (24 -> 12 -> 6 -> 3 values per lane)
        float part[24];
#if MSDA_LW_PRE
        int pk;
        float oax, oay, owg;
        MSDA_LW_OWNER(l, pk, oax, oay, owg)
#pragma unroll
        for (int p = 0; p < 8; ++p) MSDA_LW_POINT_PRE(p, pk, oax, oay, owg, part[p], part[8 + p], part[16 + p])
#else
#pragma unroll
        for (int p = 0; p < 8; ++p) MSDA_LW_POINT(l * 8 + p, part[p], part[8 + p], part[16 + p])
#endif
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const bool up = sub & (G / 2);
          const float send = up ? part[k] : part[k + 12], keep = up ? part[k + 12] : part[k];
          part[k] = keep + __shfl_xor(send, G / 2, 64);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const bool up = sub & (G / 4);
          const float send = up ? part[k] : part[k + 6], keep = up ? part[k + 6] : part[k];
          part[k] = keep + __shfl_xor(send, G / 4, 64);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool up = sub & (G / 8);
          const float send = up ? part[k] : part[k + 3], keep = up ? part[k + 3] : part[k];
          part[k] = keep + __shfl_xor(send, G / 8, 64);
        }
        if (G == 16) {
#pragma unroll
          for (int k = 0; k < 3; ++k) part[k] += __shfl_xor(part[k], 1, 64);
        }
        if constexpr (EMIT) {
          {   // the 8 d_attw sums (idx 0..7) sit three per holder lane: hand idx j to lane j of the group (one register per level)
            const int src = (sub & ~(G - 1)) + (sub < 8 ? sub / 3 : 0) * (G / 8);
            const float q0 = __shfl(part[0], src, G), q1 = __shfl(part[1], src, G), q2 = __shfl(part[2], src, G);
            const int r3 = sub % 3;
            const float mine = r3 == 0 ? q0 : r3 == 1 ? q1 : q2;
            if (l == 0) ka[0] = mine; else if (l == 1) ka[1] = mine; else if (l == 2) ka[2] = mine; else ka[3] = mine;
          }
          {   // the 16 d_loc sums of the level (idx 8 + j: d/dx of point j, idx 16 + j: d/dy): lane j collects its point's pair and
              // the group stores the level's 16 offsets gradients as ONE contiguous run (32 bytes in bf16) instead of 16 two-byte stores
            const int j = sub < 8 ? sub : 0;
            const int sx = ((8 + j) / 3) * (G / 8), sy = ((16 + j) / 3) * (G / 8);
            const float x0 = __shfl(part[0], sx, G), x1 = __shfl(part[1], sx, G), x2 = __shfl(part[2], sx, G);
            const float y0 = __shfl(part[0], sy, G), y1 = __shfl(part[1], sy, G), y2 = __shfl(part[2], sy, G);
            const int rx = (8 + j) % 3, ry = (16 + j) % 3;
            const float gx = (rx == 0 ? x0 : rx == 1 ? x1 : x2) / (float)Wl, gy = (ry == 0 ? y0 : ry == 1 ? y1 : y2) / (float)Hl;
            if (live && sub < 8) {
              T* orow = (T*)em.d_off + (g_ / nH) * em.off_ld + ((long)head * 4 + l) * 16 + sub * 2;
              if constexpr (sizeof(T) == 2) *(uint32_t*)orow = (uint32_t)f2bf(gx) | ((uint32_t)f2bf(gy) << 16);
              else *(float2*)orow = make_float2(gx, gy);
            }
          }
        } else
        if (live && (G == 8 || (sub & 1) == 0)) {
          const int base = ((sub / (G / 2)) & 1) * 12 + ((sub / (G / 4)) & 1) * 6 + ((sub / (G / 8)) & 1) * 3;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int idx = base + k;
            const int which = idx >> 3, j = l * 8 + (idx & 7);
            if (which == 0) d_attw[g_ * (long)LP + j] = part[k];
            else d_loc[(g_ * (long)LP + j) * 2 + (which - 1)] = part[k];
          }
        }
      } else {
        for (int p = 0; p < P; ++p) {
          const int j = l * P + p;
          float sv, sx, sy;
          MSDA_LW_POINT(j, sv, sx, sy)
#pragma unroll
          for (int o = G / 2; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
          if (live && sub == 0) {
            d_attw[g_ * (long)LP + j] = sv;
            *(float2*)(d_loc + (g_ * (long)LP + j) * 2) = make_float2(sx, sy);
          }
        }
      }
    }
#undef MSDA_LW_POINT
#undef MSDA_LW_OWNER
#undef MSDA_LW_POINT_PRE
    if constexpr (EMIT) {
      // softmax backward over the (q, head) group's 32 points: lane sub < 8 holds d_attw of point `sub` of every level
      float aw[4], t = 0.f;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        aw[l] = sub < 8 ? ap[l * 8 + sub] : 0.f;
        t += aw[l] * (sub < 8 ? ka[l] : 0.f);
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (live && sub < 8) {
        T* lrow = (T*)em.d_logit + (g_ / nH) * em.logit_ld + (long)head * 32 + sub;
#pragma unroll
        for (int l = 0; l < 4; ++l) Io<T>::st(lrow + l * 8, aw[l] * (ka[l] - t));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// d_value without per-channel atomics: bin the sampling points by value tile, then accumulate each tile in REGISTERS.
//
// Measured on MI355X (tools/ubench/ubench): the L2 executes fp32 atomics at ~1 dword/clock/channel (5.0 G 256-byte bursts/s
// chip-wide, independent of footprint and scope) and LDS fp32 atomics are slower still (ds_add_f32: ~170 cycles per
// wave-instruction per CU).  The direct scatter needs 4 x 64 dword atomics per sampling point, which pins the
// cross-attention of 8 images at ~140 ms.  Binning needs ONE integer (LDS) atomic per record:
//   count : one sampling point per lane (perfectly coalesced loc / attw) -> LDS histogram of the tiles it touches
//   scan  : exclusive prefix of the counts -> bin offsets, and the list of 4096-record chunks
//   fill  : same traversal, slot from the LDS cursor; records[slot] = {query, corner position in the tile, w, ax, ay}
//   drain : persistent waves, one chunk each; see msda_drain_k.
// A bin = one 8 x 4 TILE of positions of one (batch, head, level).  The four bilinear corners of a sampling point share
// ONE gradient row, and with a 2-D tile they share one tile 71 % of the time: a record is a (point, tile) pair — 1.41 per
// point on average instead of 4 (point, corner) entries — so the drain gathers 2.8x fewer gradient rows and the record
// list is 22.6 instead of 32 bytes per point.
// (tile geometry, MsdaBins and MsdaWs: msda.h — shared with the MFMA drain in msda_drain_mfma.hip)

// Binning = a counting sort with workgroup-private LDS histograms (global integer atomics cost one L2 request per
// lane: 8e8 of them took 45-70 ms; LDS counters are private to the CU).  A work unit = (image, head, one of R contiguous
// ranges of the queries); a workgroup of 1024 threads walks its unit's sampling points, one per lane per iteration (the
// 32 points of a (query, head) are 256 contiguous bytes of loc).
//   COUNT: hist[tile]++ in LDS, then the histogram is stored to seg_hist[unit][.] and added to cnt[bin]
//   FILL : LDS cursors start at the unit's exclusive offsets (seg_hist rewritten in place by msda_segscan_k) and
//          hand out slots: records[slot] = {query << 7 | (ly+1) << 4 | (lx+1), weight, frac x, frac y}
// Why few, large units (round 3; before: 65536-point segments over all heads, ~1000 workgroups resident): every
// (unit, bin) pair owns a contiguous run of record slots that the unit fills 16 bytes at a time, so the set of partially
// written 128-byte lines is (#resident units x #bins of a unit).  With 1000 units x 8800 bins that was 1.1 GB — every line left
// the caches several times before it was complete and the fill pass ran at the speed of partial-line writes to HBM: 3.56 ms for the
// cross-attention against 1.36 ms with the stores removed and 1.67 ms with the same bytes stored linearly (scratch experiment,
// DESIGN.md §5).  One unit per CU and one head per unit leaves 256 x 1100 lines = 36 MB, which the L2s + the 256 MB Infinity Cache hold.
template <bool FILL>
__global__ void __launch_bounds__(1024) msda_hist_k(MsdaLevels lv, MsdaBins bins, const float* __restrict__ loc,
                                                    const float* __restrict__ attw, MsdaWs ws, int Nq, int nH, int L, int P, int R, int B) {
  extern __shared__ int hist[];
  const int ntiles = bins.first_tile[L];
  const int LP = L * P;
  const int nunits = B * nH * R;
  // per-point index arithmetic in 32 bits with shifts when L*P and P are powers of two (every GEDepth config): a division by a
  // run-time value is ~50 VALU instructions, and this loop has one thread per sampling point
  const int sh_lp = (LP & (LP - 1)) ? -1 : __ffs(LP) - 1, sh_p = (P & (P - 1)) ? -1 : __ffs(P) - 1;
  for (int u = blockIdx.x; u < nunits; u += gridDim.x) {          // u = (b * nH + head) * R + r
    const int r = u % R, bh = u / R;
    const int head = bh % nH, b = bh / nH;
    int* gh = ws.seg_hist + (long)u * ntiles;
    for (int i = threadIdx.x; i < ntiles; i += 1024) hist[i] = FILL ? gh[i] : 0;
    __syncthreads();
    const int q_lo = (int)((long)Nq * r / R), q_hi = (int)((long)Nq * (r + 1) / R);
    const int n = (q_hi - q_lo) * LP;                              // launcher: (Nq / R + 1) * LP < 2^31
    const long pt0 = ((long)b * Nq + q_lo) * nH + head;            // (b, q, head) group of the unit's first query
    for (int j = threadIdx.x; j < n; j += 1024) {
      const int qi = sh_lp >= 0 ? (j >> sh_lp) : j / LP;
      const int lp = j - mul24(qi, LP);
      const int l = sh_p >= 0 ? (lp >> sh_p) : lp / P;
      const int q = q_lo + qi;
      const long pt = (pt0 + (long)qi * nH) * LP + lp;
      const int Hl = lv.H[l], Wl = lv.W[l];
      const float2 xy = ((const float2*)loc)[pt];
      const float x = xy.x * (float)Wl - 0.5f, y = xy.y * (float)Hl - 0.5f;
      if (!(y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl)) continue;
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float ax = x - xf, ay = y - yf;
      const float wgt = FILL ? attw[pt] : 0.f;
      const int ntx = bins.ntx[l];
      const int lb0 = bins.first_tile[l];
      // tile columns / rows that hold an in-image corner: the left column x0 (if >= 0) and the right column x0+1 (if < W)
      const bool xa = x0 >= 0, xb = x0 + 1 < Wl, ya = y0 >= 0, yb = y0 + 1 < Hl;
      const int txa = x0 >> 3, txb = (x0 + 1) >> 3, tya = y0 >> 2, tyb = (y0 + 1) >> 2;
      const int tx_first = xa ? txa : txb, ty_first = ya ? tya : tyb;
      const int two_x = (xa && xb && txb != txa) ? 1 : 0, two_y = (ya && yb && tyb != tya) ? 1 : 0;
      for (int jy = 0; jy <= two_y; ++jy)
        for (int jx = 0; jx <= two_x; ++jx) {
          const int tx = jx ? txb : tx_first, ty = jy ? tyb : ty_first;
          const int slot = atomicAdd(&hist[lb0 + mul24(ty, ntx) + tx], 1);        // LDS
          if (FILL) {
            const int lx1 = x0 - tx * MSDA_TW + 1, ly1 = y0 - ty * MSDA_TH + 1;   // top-left corner relative to the tile, +1: [0,8] x [0,4]
            ws.entries[slot] = make_int4((q << 7) | (ly1 << 4) | lx1, __float_as_int(wgt), __float_as_int(ax), __float_as_int(ay));
          }
        }
    }
    __syncthreads();
    if (!FILL) {
      for (int i = threadIdx.x; i < ntiles; i += 1024) {
        const int c = hist[i];
        gh[i] = c;
        if (c) atomicAdd(&ws.cnt[bh * ntiles + i], c);
      }
      __syncthreads();
    }
  }
}

// The same two passes fed from the RAW projection outputs (bf16) + reference points instead of the fp32 loc / attw tensors (round 4:
// those 2.4 GB per cross-attention launch are no longer written by the forward, ge_msda_fwd_mm), L == 4, P == 8: thread = sampling
// point, the 32 points of a (query, head) are one half wave — locations in the forward's arithmetic to the bit (mm_div: the tile and
// the cell of every tap must be the forward's), the softmax over the half wave on the DPP network (FILL only), 8-byte records.
struct MsdaRawIn {
  const bf16_t* off; long off_ld; const bf16_t* logit; long logit_ld;
  const float* ref; long ref_sb, ref_sq, ref_sl;
  float fW[4], fH[4], rW[4], rH[4];                     // map sizes and their correctly rounded reciprocals
};
__device__ __forceinline__ float msda_div(float n, float d, float r) {       // = n / d for bf16-valued n, integer d <= 8191 (msda_mm.hip)
  const float q0 = n * r;
  return __builtin_fmaf(__builtin_fmaf(-q0, d, n), r, q0);
}
__device__ __forceinline__ float msda_half_max(float v) {                    // over the 32 lanes of a half wave, result in every lane
#define MSDA_DPP_F(V, CTRL, RM) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(V), __float_as_int(V), CTRL, RM, 0xf, false))
  v = fmaxf(v, MSDA_DPP_F(v, 0x111, 0xf)); v = fmaxf(v, MSDA_DPP_F(v, 0x112, 0xf));
  v = fmaxf(v, MSDA_DPP_F(v, 0x114, 0xf)); v = fmaxf(v, MSDA_DPP_F(v, 0x118, 0xf));
  v = fmaxf(v, MSDA_DPP_F(v, 0x142, 0xa));
#undef MSDA_DPP_F
  const float a = readlane_f(v, 31), b = readlane_f(v, 63);
  return (threadIdx.x & 32) ? b : a;
}
__device__ __forceinline__ float msda_half_sum(float v) {
#define MSDA_DPP_Z(V, CTRL, RM) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, RM, 0xf, true))
  v += MSDA_DPP_Z(v, 0x111, 0xf); v += MSDA_DPP_Z(v, 0x112, 0xf);            // inclusive scan inside the 16-lane rows
  v += MSDA_DPP_Z(v, 0x114, 0xf); v += MSDA_DPP_Z(v, 0x118, 0xf);
  v += MSDA_DPP_Z(v, 0x142, 0xa);                                            // row totals of rows 0 / 2 into rows 1 / 3
#undef MSDA_DPP_Z
  const float a = readlane_f(v, 31), b = readlane_f(v, 63);
  return (threadIdx.x & 32) ? b : a;
}
template <bool FILL>
__global__ void __launch_bounds__(1024) msda_hist_raw_k(MsdaLevels lv, MsdaBins bins, MsdaRawIn in, MsdaWs ws, int Nq, int nH, int R, int B,
                                                        int level_mask) {
  extern __shared__ int hist[];
  const int ntiles = bins.first_tile[4];
  const int nunits = B * nH * R;
  for (int u = blockIdx.x; u < nunits; u += gridDim.x) {          // u = (b * nH + head) * R + r
    const int r = u % R, bh = u / R;
    const int head = bh % nH, b = bh / nH;
    int* gh = ws.seg_hist + (long)u * ntiles;
    for (int i = threadIdx.x; i < ntiles; i += 1024) hist[i] = FILL ? gh[i] : 0;
    __syncthreads();
    const int q_lo = (int)((long)Nq * r / R), q_hi = (int)((long)Nq * (r + 1) / R);
    const int n = (q_hi - q_lo) * 32;
    for (int j0 = 0; j0 < n; j0 += 1024) {                        // uniform trip count per wave: the softmax needs whole half waves
      const int j = j0 + threadIdx.x;
      const bool live = j < n;
      const int qi = live ? j >> 5 : 0, lp = j & 31, l = lp >> 3;
      const int q = q_lo + qi;
      const long row = (long)b * Nq + q;
      const uint32_t o = *(const uint32_t*)(in.off + row * in.off_ld + head * 64 + lp * 2);
      const float* rp = in.ref + (long)b * in.ref_sb + (long)q * in.ref_sq + (long)l * in.ref_sl;
      const float fW = in.fW[l], fH = in.fH[l];
      const float lx = rp[0] + msda_div(__uint_as_float(o << 16), fW, in.rW[l]);
      const float ly = rp[1] + msda_div(__uint_as_float(o & 0xffff0000u), fH, in.rH[l]);
      const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
      float wgt = 0.f;
      if (FILL) {
        const float lg = bf2f(in.logit[row * in.logit_ld + head * 32 + lp]);
        const float e = __expf(lg - msda_half_max(lg));
        wgt = e * (1.f / msda_half_sum(e));
      }
      // level_mask: levels whose d_value another kernel produces (ge_msda_bwd_value_mm) leave no records here
      if (!(live && ((level_mask >> l) & 1) && y > -1.f && x > -1.f && y < fH && x < fW)) continue;
      const int Hl = lv.H[l], Wl = lv.W[l];
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float ax = x - xf, ay = y - yf;
      const int ntx = bins.ntx[l];
      const int lb0 = bins.first_tile[l];
      const bool xa = x0 >= 0, xb = x0 + 1 < Wl, ya = y0 >= 0, yb = y0 + 1 < Hl;
      const int txa = x0 >> 3, txb = (x0 + 1) >> 3, tya = y0 >> 2, tyb = (y0 + 1) >> 2;
      const int tx_first = xa ? txa : txb, ty_first = ya ? tya : tyb;
      const int two_x = (xa && xb && txb != txa) ? 1 : 0, two_y = (ya && yb && tyb != tya) ? 1 : 0;
      // bf16 weight, fractions at 8 bits (floor: the drain reads them back at the centre of the step)
      const uint32_t packed = (uint32_t)f2bf(wgt) | ((uint32_t)min((int)(ax * 256.f), 255) << 16) | ((uint32_t)min((int)(ay * 256.f), 255) << 24);
      for (int jy = 0; jy <= two_y; ++jy)
        for (int jx = 0; jx <= two_x; ++jx) {
          const int tx = jx ? txb : tx_first, ty = jy ? tyb : ty_first;
          const int slot = atomicAdd(&hist[lb0 + mul24(ty, ntx) + tx], 1);        // LDS
          if (FILL) {
            const int lx1 = x0 - tx * MSDA_TW + 1, ly1 = y0 - ty * MSDA_TH + 1;
            ((int2*)ws.entries)[slot] = make_int2((q << 7) | (ly1 << 4) | lx1, (int)packed);
          }
        }
    }
    __syncthreads();
    if (!FILL) {
      for (int i = threadIdx.x; i < ntiles; i += 1024) {
        const int c = hist[i];
        gh[i] = c;
        if (c) atomicAdd(&ws.cnt[bh * ntiles + i], c);
      }
      __syncthreads();
    }
  }
}

// seg_hist[unit = bh * R + r][tile] (counts) -> absolute first slot of unit r in bin bh * ntiles + tile
__global__ void __launch_bounds__(256) msda_segscan_k(MsdaWs ws, int ntiles, int R, int nbins) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbins) return;
  const int bh = i / ntiles, tile = i - bh * ntiles;
  int run = (int)ws.offset[i];
  int* p = ws.seg_hist + (long)bh * R * ntiles + tile;
  for (int r = 0; r < R; ++r) { const int c = p[(long)r * ntiles]; p[(long)r * ntiles] = run; run += c; }
}

// exclusive scans of the bin counts (-> first record slot) and of the bins' chunk counts (-> first chunk index).  One workgroup of 16
// waves; wave w owns the contiguous segment [w * seg, (w + 1) * seg) and walks it 64 bins at a time (coalesced), scanning each batch with
// lane shuffles: pass 1 totals per wave, 16-entry scan, pass 2 writes.  (The first version ran a 1024-wide Hillis-Steele scan with 20
// barriers for every 1024 bins: 170 us for 70 k bins; this one is a few microseconds.)
__device__ __forceinline__ void msda_wave_scan(long& vo, int& vk, int lane) {      // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long o = __shfl_up(vo, d, 64);
    const int k = __shfl_up(vk, d, 64);
    if (lane >= d) { vo += o; vk += k; }
  }
}
__global__ void __launch_bounds__(1024) msda_scan_k(MsdaWs ws, int nbins) {
  // EIGHT bins per lane and step (two 16-byte loads; local prefix, then the wave scan over the lane totals): the loop is a chain of dependent
  // global round trips, and with one bin per lane a 65 k-bin scan took 128 of them = 86 us per launch (rocprofv3, round 4)
  __shared__ long w_off[16];
  __shared__ int w_chk[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int seg = ((nbins + 15) / 16 + 511) / 512 * 512;
  const int lo = wv * seg, hi = min(nbins, lo + seg);
  auto load8 = [&](int i0, int* c) {
    if (i0 + 8 <= hi) {
      const int4 a = *(const int4*)(ws.cnt + i0), b = *(const int4*)(ws.cnt + i0 + 4);
      c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) c[e] = i0 + e < hi ? ws.cnt[i0 + e] : 0;
    }
  };
  long to = 0; int tk = 0;
  for (int base = lo; base < hi; base += 512) {
    int c[8];
    load8(base + lane * 8, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) { to += c[e]; tk += (c[e] + MSDA_CHUNK - 1) / MSDA_CHUNK; }
  }
#pragma unroll
  for (int d = 32; d; d >>= 1) { to += __shfl_xor(to, d, 64); tk += __shfl_xor(tk, d, 64); }
  if (lane == 0) { w_off[wv] = to; w_chk[wv] = tk; }
  __syncthreads();
  long run_o = 0; int run_k = 0;
  for (int w = 0; w < wv; ++w) { run_o += w_off[w]; run_k += w_chk[w]; }
  for (int base = lo; base < hi; base += 512) {
    const int i0 = base + lane * 8;
    int c[8], kc[8];
    load8(i0, c);
    long vo = 0; int vk = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { kc[e] = (c[e] + MSDA_CHUNK - 1) / MSDA_CHUNK; vo += c[e]; vk += kc[e]; }
    long so = vo; int sk = vk;
    msda_wave_scan(so, sk, lane);
    long o = run_o + so - vo; int k = run_k + sk - vk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (i0 + e < hi) { ws.offset[i0 + e] = o; ws.chunk_first[i0 + e] = k; }
      o += c[e]; k += kc[e];
    }
    run_o += __shfl(so, 63, 64);
    run_k += __shfl(sk, 63, 64);
  }
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < 16; ++w) total += w_chk[w];
    ws.chunk_first[nbins] = total; ws.ctrl[0] = total;
  }
  if (threadIdx.x >= 1 && threadIdx.x < 2 + MSDA_XCDS) ws.ctrl[threadIdx.x] = 0;      // per-XCD work cursors
}

// Work order of the drain.  The records of a bin are laid out unit by unit, i.e. sorted by the QUERY RANGE r they come from, and a chunk
// is MSDA_CHUNK consecutive records of one bin.  Drained in bin order — (image, head, level, tile) — the waves in flight on an XCD (3
// workgroups x 4 waves x 32 CUs = 384 chunks) walk a whole level of the (image, head) at a time, and their records point at gradient
// rows all over the 12.6 MB of d_out rows that (image, head) has (KITTI cross-attention) — three times the XCD's L2, and every level
// fetches them again: 6.8 GB of re-fetched rows per launch (rocprofv3 FETCH_SIZE, profiles/pmc_traffic.json) for 0.8 GB of d_out.
// Here the chunks of an (image, head) are listed range by range instead: item -> chunk id with all chunks whose FIRST record comes from
// query range 0 first (tiles of all four levels), then range 1, ...  What is in flight then reads the d_out rows of one or two ranges
// (Nq / R rows x 128 B = 1.6 MB each at R = 8), which stay in L2 across the levels.  One workgroup per (image, head): count the
// chunks per range in LDS, exclusive scan, place.  The order inside a range is the order the LDS atomics resolve in (any order is correct:
// chunks of one bin meet in d_value through fp32 atomics either way).
#define MSDA_MAX_R 512
__global__ void __launch_bounds__(1024) msda_order_k(MsdaWs ws, int ntiles, int R, int grouped) {
  __shared__ int s_cnt[MSDA_MAX_R], s_base[MSDA_MAX_R];
  const int bh = blockIdx.x;
  for (int i = threadIdx.x; i < R; i += 1024) s_cnt[i] = 0;
  __syncthreads();
  const int* seg = ws.seg_hist + (long)bh * R * ntiles;          // [r][tile]: absolute first slot of range r in the bin (after msda_segscan_k)
  const int first = ws.chunk_first[bh * ntiles];
  if (!grouped) {                                                 // bin order (mode bit 6 off: A/B runs)
    const int last = ws.chunk_first[(bh + 1) * ntiles];
    for (int i = first + threadIdx.x; i < last; i += 1024) ws.order[i] = i;
    return;
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (int tile = threadIdx.x; tile < ntiles; tile += 1024) {
      const int bin = bh * ntiles + tile;
      const int c0 = ws.chunk_first[bin], nch = ws.chunk_first[bin + 1] - c0;
      const long off0 = ws.offset[bin];
      int r = 0;
      for (int c = 0; c < nch; ++c) {
        const long slot = off0 + (long)c * MSDA_CHUNK;
        while (r + 1 < R && seg[(long)(r + 1) * ntiles + tile] <= slot) ++r;        // range of the chunk's first record
        const int k = atomicAdd(&s_cnt[r], 1);
        if (pass) ws.order[first + s_base[r] + k] = c0 + c;
      }
    }
    __syncthreads();
    if (pass == 0) {
      if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < R; ++i) { s_base[i] = run; run += s_cnt[i]; s_cnt[i] = 0; } }
      __syncthreads();
    }
  }
}

// drain: a wave owns one chunk (<= MSDA_CHUNK records of one tile) and accumulates the tile in 32 VGPRs per lane
// (lane == channel).  History (8 x 352 x 1120 cross-attention): v1 consumed (point, corner) entries in arrival order with
// `acc[pos] += g * coef`, pos a dynamic register index: 6 VALU + 5 SALU per entry, VALU-issue bound, 12.5 ms.  v2 sorted
// each sub-chunk by position in LDS and consumed run by run: 3 VALU per entry, then bound by the 4 gradient-row gathers
// per sampling point.  v3 (this one) works on (point, tile) records: one gather per record, sub-chunks counting-sorted in
// LDS by the corner class (ly+1)*9 + (lx+1) — lane-parallel, ~0.3 instructions per record — and consumed run by run into
// four statically addressed accumulators (the 2 x 2 corners) with two packed FMAs per record.  Classes are visited in
// raster order, so the right-hand pair of one class is the left-hand pair of the next: per class step two accumulators
// are written to the register tile through `s_set_gpr_idx` and two are carried over.
#define MSDA_SUB 512           // measured 256 / 512 / 768: 5.0 / 4.3 / 5.2 ms (LDS per wave sets the occupancy, class runs get shorter)
#define MSDA_NCLS 45           // (ly+1) in [0,4] x (lx+1) in [0,8]
template <typename T>
__global__ void __launch_bounds__(256) msda_drain_k(MsdaLevels lv, MsdaBins bins, MsdaWs ws, const T* __restrict__ gout,
                                                    float* __restrict__ d_value, int nbins, int Nv, int Nq, int nH, int L) {
  constexpr int CPLr = 16 / (int)sizeof(T);            // channels per lane in a 16-byte gather
  constexpr int LPR = 64 / CPLr;                       // lanes per gradient row: 8 (bf16) / 16 (fp32)
  constexpr int RPI = 64 / LPR;                        // rows per gather instruction: 8 / 4
  constexpr int NR = 32 / RPI;                         // gather instructions per 32-row block: 4 / 8
  __shared__ __attribute__((aligned(16))) T stage_all[4 * 32 * 64];   // per wave: 32 rows x 64 channels
  __shared__ int s_key_all[4 * MSDA_SUB];
  __shared__ float s_w_all[4 * MSDA_SUB], s_ax_all[4 * MSDA_SUB], s_ay_all[4 * MSDA_SUB];
  __shared__ int s_start_all[4 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* stage = stage_all + (size_t)wv * (32 * 64);
  int* s_key = s_key_all + wv * MSDA_SUB;
  float* s_w = s_w_all + wv * MSDA_SUB;
  float* s_ax = s_ax_all + wv * MSDA_SUB;
  float* s_ay = s_ay_all + wv * MSDA_SUB;
  int* s_start = s_start_all + wv * 64;                // [0..44] run starts, [45] = n, [46] = never reached
  const int gi = lane / LPR, subc = (lane % LPR) * CPLr;
  const int ntiles = bins.first_tile[L];
  const int total = ws.ctrl[0];
  // chunks are ordered by (image, head, tile): XCD x drains the x-th eighth of the list, so that neighbouring tiles — whose
  // records point at the same gradient rows — meet in one L2; a wave that runs dry helps the next partition
  const int part0 = blockIdx.x % MSDA_XCDS;
  int probe = 0;
  for (;;) {
    int item = -1;
    while (probe < MSDA_XCDS) {
      const int part = (part0 + probe) % MSDA_XCDS;
      const int p_lo = (int)((long)total * part / MSDA_XCDS), p_hi = (int)((long)total * (part + 1) / MSDA_XCDS);
      int k = 0;
      if (lane == 0) k = atomicAdd(&ws.ctrl[2 + part], 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (p_lo + k < p_hi) { item = p_lo + k; break; }
      ++probe;
    }
    if (item < 0) break;
    if (ws.order) item = ws.order[item];                  // work order -> chunk id (msda_order_k, mode bit 6)
    int lo = 0, hi = nbins;                               // largest bin with chunk_first[bin] <= item
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ws.chunk_first[mid] <= item) lo = mid; else hi = mid; }
    const int bin = lo;
    const int nchunk = ws.chunk_first[bin + 1] - ws.chunk_first[bin];
    const int chunk = item - ws.chunk_first[bin];
    const int cnt = ws.cnt[bin];
    const int e_lo = chunk * MSDA_CHUNK, e_hi = min(cnt, e_lo + MSDA_CHUNK);
    const int4* ent = ws.entries + ws.offset[bin];
    const int bh = bin / ntiles, tile = bin - bh * ntiles;
    const int b = bh / nH, head = bh - b * nH;
    int l = 0;
    while (l + 1 < L && tile >= bins.first_tile[l + 1]) ++l;
    const int tl = tile - bins.first_tile[l];
    const int ty = tl / bins.ntx[l], tx = tl - ty * bins.ntx[l];
    const long rowbase = (long)b * Nq * nH + head;

    f32x32_t a0 = 0.f;
    for (int s_lo = e_lo; s_lo < e_hi; s_lo += MSDA_SUB) {
      const int n = min(MSDA_SUB, e_hi - s_lo);
      // ---- counting sort of the sub-chunk by corner class (one record per lane per step; LDS integer atomics are cheap)
      __builtin_amdgcn_wave_barrier();
      s_start[lane] = 0;
      __builtin_amdgcn_wave_barrier();
      int4 E[MSDA_SUB / 64];
      int rank[MSDA_SUB / 64];
#pragma unroll
      for (int i = 0; i < MSDA_SUB / 64; ++i) {
        const int idx = i * 64 + lane;
        E[i] = make_int4(0, 0, 0, 0);
        rank[i] = 0;
        if (idx < n) {
          E[i] = ent[s_lo + idx];
          rank[i] = atomicAdd(&s_start[((E[i].x >> 4) & 7) * 9 + (E[i].x & 15)], 1);
        }
      }
      __builtin_amdgcn_wave_barrier();
      {
        const int c = lane < MSDA_NCLS ? s_start[lane] : 0;
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        __builtin_amdgcn_wave_barrier();
        s_start[lane] = lane < MSDA_NCLS ? inc - c : (lane == MSDA_NCLS ? n : 0x7fffffff);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < MSDA_SUB / 64; ++i) {
        const int idx = i * 64 + lane;
        if (idx < n) {
          const int slot = s_start[((E[i].x >> 4) & 7) * 9 + (E[i].x & 15)] + rank[i];
          s_key[slot] = E[i].x;
          s_w[slot] = __int_as_float(E[i].y);
          s_ax[slot] = __int_as_float(E[i].z);
          s_ay[slot] = __int_as_float(E[i].w);
        }
      }
      const int nb = (n + 31) >> 5;                        // 32-record blocks; pad the last one with weight 0 / query 0
      if (n + lane < nb * 32 && lane < 32) { s_key[n + lane] = 0; s_w[n + lane] = 0.f; s_ax[n + lane] = 0.f; s_ay[n + lane] = 0.f; }
      __builtin_amdgcn_wave_barrier();

      // ---- gather (next block in flight while this one is consumed) + run-wise accumulation
      u32x4_t R[NR];                                       // (HIP's uint4 struct kept this array in scratch)
#define MSDA_GATHER(BLK)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                                         \
    const int q = s_key[(BLK) * 32 + i * RPI + gi] >> 7;                                                   \
    R[i] = *(const u32x4_t*)(gout + (rowbase + (long)q * nH) * 64 + subc);                                 \
  }
#define MSDA_PARK()                                                                                        \
  _Pragma("unroll") for (int i = 0; i < NR; ++i) *(u32x4_t*)(stage + (i * RPI + gi) * 64 + subc) = R[i];
      // a0[p] += v for a wave-uniform p, only when `ok` (a branch around it makes the compiler copy all 32 registers)
#define MSDA_PUT(P_, OK_, V_) a0[(P_) & 31] += (OK_) ? (V_) : 0.f
      int cls = 0;
      int run_end = __builtin_amdgcn_readfirstlane(s_start[1]);
      f32x2_t accT = {0.f, 0.f}, accB = {0.f, 0.f};        // top pair (ly: lx, lx+1), bottom pair (ly+1: lx, lx+1)
      MSDA_GATHER(0)
      MSDA_PARK()
      for (int blk = 0; blk < nb; ++blk) {
        const int i0 = blk * 32;
        // the 32 gradient values of this lane's channel are fetched up front: the run-boundary branches below would
        // otherwise serialise every record on an LDS round trip; then the stage is free for the next block
        float gr[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) gr[k] = Ld1<T>::ld(stage + k * 64 + lane);
        if (blk + 1 < nb) { MSDA_GATHER(blk + 1) }
        // bilinear corner coefficients of record i0 + (lane & 31), zero for the corners that lie outside this tile
        f32x2_t cT, cB;
        {
          const int j = i0 + (lane & 31);
          const int key = s_key[j];
          const float w = s_w[j], ax = s_ax[j], ay = s_ay[j];
          const int lx = (key & 15) - 1, ly = ((key >> 4) & 7) - 1;
          const float wl = lx >= 0 ? 1.f - ax : 0.f, wr = lx < MSDA_TW - 1 ? ax : 0.f;
          const float wt = ly >= 0 ? w * (1.f - ay) : 0.f, wb = ly < MSDA_TH - 1 ? w * ay : 0.f;
          cT = f32x2_t{wt * wl, wt * wr};
          cB = f32x2_t{wb * wl, wb * wr};
        }
        // the wave's own instruction stream is what this kernel is bound by: the run boundary is tested against
        // a block-relative value (compare with a constant), and the four wave-uniform coefficients of record k + 1 are read
        // while record k is accumulated, so that no hazard nops separate a v_readlane from the FMA that uses its SGPR
        int rel = run_end - i0;
        float nTx = readlane_f(cT.x, 0), nTy = readlane_f(cT.y, 0), nBx = readlane_f(cB.x, 0), nBy = readlane_f(cB.y, 0);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          if (__builtin_expect(k == rel, 0)) {               // wave-uniform and rare: keep the hot path fall-through
            do {                                             // leave class `cls` (and any empty classes behind it)
              const int ly = cls / 9 - 1, lx = cls % 9 - 1;
              const bool in_cls = cls < MSDA_NCLS;
              MSDA_PUT(ly * MSDA_TW + lx, in_cls && ly >= 0 && lx >= 0, accT.x);
              MSDA_PUT((ly + 1) * MSDA_TW + lx, in_cls && ly < MSDA_TH - 1 && lx >= 0, accB.x);
              // raster order: the right-hand pair becomes the left-hand pair of the next class of this tile row
              const bool carry = lx < MSDA_TW - 1;
              accT = f32x2_t{carry ? accT.y : 0.f, 0.f};
              accB = f32x2_t{carry ? accB.y : 0.f, 0.f};
              ++cls;
              run_end = __builtin_amdgcn_readfirstlane(s_start[min(cls + 1, MSDA_NCLS + 1)]);
              rel = run_end - i0;
            } while (k == rel);
          }
          const f32x2_t kT = {nTx, nTy}, kB = {nBx, nBy};
          if (k + 1 < 32) {
            nTx = readlane_f(cT.x, k + 1); nTy = readlane_f(cT.y, k + 1); nBx = readlane_f(cB.x, k + 1); nBy = readlane_f(cB.y, k + 1);
          }
          const float g = gr[k];
          const f32x2_t g2 = {g, g};
          accT += kT * g2;
          accB += kB * g2;
        }
        if (blk + 1 < nb) { MSDA_PARK() }
      }
      {                                                      // close the last open class: all four corners
        const int ly = cls / 9 - 1, lx = cls % 9 - 1;
        const bool in_cls = cls < MSDA_NCLS;
        MSDA_PUT(ly * MSDA_TW + lx, in_cls && ly >= 0 && lx >= 0, accT.x);
        MSDA_PUT((ly + 1) * MSDA_TW + lx, in_cls && ly < MSDA_TH - 1 && lx >= 0, accB.x);
        MSDA_PUT(ly * MSDA_TW + lx + 1, in_cls && ly >= 0 && lx < MSDA_TW - 1, accT.y);
        MSDA_PUT((ly + 1) * MSDA_TW + lx + 1, in_cls && ly < MSDA_TH - 1 && lx < MSDA_TW - 1, accB.y);
      }
#undef MSDA_GATHER
#undef MSDA_PARK
#undef MSDA_PUT
    }
    // tile -> d_value rows (y, x) of level l; positions of a border tile that fall outside the map were never real
    const int Wl = lv.W[l], Hl = lv.H[l];
    float* dst = d_value + (((long)b * Nv + lv.start[l]) * nH + head) * 64 + lane;
    const long pstride = (long)nH * 64;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int y = ty * MSDA_TH + e / MSDA_TW, x = tx * MSDA_TW + e % MSDA_TW;
      if (y < Hl && x < Wl) {
        float* p = dst + ((long)y * Wl + x) * pstride;
        if (nchunk == 1) *p = a0[e];
        else atomicAdd(p, a0[e]);
      }
    }
  }
}

static size_t msda_ws_layout(int nbins, long seg_hist_ints, long max_entries, char* base, MsdaWs* ws) {
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_cnt = carve((size_t)nbins * 4), o_chk = carve((size_t)(nbins + 1) * 4);
  const size_t o_off = carve((size_t)(nbins + 1) * 8), o_ctrl = carve(64), o_sh = carve((size_t)seg_hist_ints * 4);
  const size_t o_ord = carve((size_t)(nbins + max_entries / MSDA_CHUNK + 1) * 4);       // chunks <= bins + records / MSDA_CHUNK
  const size_t o_ent = carve((size_t)max_entries * 16);
  if (ws) {
    ws->order = (int*)(base + o_ord);
    ws->cnt = (int*)(base + o_cnt); ws->chunk_first = (int*)(base + o_chk); ws->offset = (long*)(base + o_off);
    ws->ctrl = (int*)(base + o_ctrl); ws->seg_hist = (int*)(base + o_sh); ws->entries = (int4*)(base + o_ent);
  }
  return off;
}
struct MsdaPlan { int ntiles, nloc, R, nbins; long max_entries; bool ok; };
// Work units of the counting sort = resident 1024-thread workgroups: two per CU (2048 threads, the CU's limit).  Measured, cross /
// self launch (ms): 128 units count 1.56 / 0.53, fill 3.54 / 1.15; 256: 0.81 / 0.28, 2.47 / 0.73; 512: 0.55 / 0.19, 2.17 / 0.73;
// 1024 = 512 (only 512 are resident).  The passes wait on memory latency, so threads in flight are what counts.
#ifndef MSDA_HIST_WGS
#define MSDA_HIST_WGS 512
#endif
static MsdaPlan msda_plan(const MsdaBins& bins, int B, int Nq, int nH, int L, int P) {
  MsdaPlan pl;
  pl.ntiles = bins.first_tile[L];
  pl.nloc = nH * pl.ntiles;
  const long npts_b = (long)Nq * nH * L * P;
  // query ranges per (image, head): enough units for one per CU, each with at least a few thousand points
  long R = (MSDA_HIST_WGS + (long)B * nH - 1) / std::max(1L, (long)B * nH);
  R = std::max(1L, std::min(R, (long)std::max(1, Nq / 64)));
  pl.R = (int)R;
  pl.max_entries = (long)B * npts_b * 4;
  const long nbins = (long)B * pl.nloc;
  pl.nbins = (int)nbins;
  const size_t hist_bytes = (size_t)pl.ntiles * 4;
  pl.ok = nbins < (1L << 30) && Nq < (1 << 24) && pl.max_entries < (1L << 31) && hist_bytes <= 60 * 1024 && B <= 65535 &&
          ((long)Nq / R + 1) * L * P < (1L << 31) && (long)B * nH * R * pl.ntiles < (1L << 31) && R <= MSDA_MAX_R;
  return pl;
}
static int msda_bins(const MsdaLevels& lv, int L, MsdaBins& bins) {
  int n = 0;
  for (int l = 0; l < L; ++l) {
    bins.first_tile[l] = n;
    bins.ntx[l] = (lv.W[l] + MSDA_TW - 1) / MSDA_TW;
    n += bins.ntx[l] * ((lv.H[l] + MSDA_TH - 1) / MSDA_TH);
  }
  bins.first_tile[L] = n;
  return n;
}


// Kernel selection: bit 0 = LDS-window forward, bit 1 = LDS-window d_loc / d_attw (both need the query geometry), bit 2 =
// owner-lane tap arithmetic in the window kernels, bit 3 = head-major work order in the streaming kernels; the
// streaming kernels serve everything else.  A process-wide knob for A/B timing and for the tests that compare the two.
// bit 4 = bf16 d_value drain on the matrix cores (msda_drain_mfma.hip), bit 5 = its B operand through ds_read_b64_tr_b16,
// bit 6 = drain work order grouped by query range (msda_order_k) instead of bin order: opt-in — it cuts the drain's re-fetch traffic by
// 28 % but the ordering pass costs what the drain gains (profiles/r4_ab_drain_order.txt).
static int g_msda_mode = 13 | 16 | 32;   // window forward + owner-lane taps + head-major streaming d_loc/d_attw + MFMA drain (measured best); bit 6 (range-grouped drain order) is opt-in
extern "C" int ge_msda_mode(int mode) {
  const int old = g_msda_mode;
  if (mode >= 0) g_msda_mode = mode & 127;
  return old;
}

extern "C" int ge_msda_fwd(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc,
                           const float* attw, void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !loc || !attw || !out || B < 0 || Nv <= 0 || Nq < 0 || nH <= 0 || P <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long n_groups = (long)B * Nq * nH;
  if (n_groups == 0) return GE_OK;
  if (dtype != GE_F32 && dtype != GE_BF16) return GE_ERR_UNSUPPORTED;
  if ((g_msda_mode & 1) && query_hw && n_qseg > 0 && msda_win_supported(B, Nq, nH, L, P, Nv))
    return msda_fwd_win_launch(value, lv, query_hw, n_qseg, loc, attw, out, B, Nv, Nq, nH, L, P, dtype, (g_msda_mode & 4) != 0, ge_stream(stream));
  if ((n_groups + 15) / 16 > (1L << 30) || (long)Nv * nH * 64 >= (1L << 31)) return GE_ERR_UNSUPPORTED;
  const unsigned blocks = msda_grid(n_groups, dtype == GE_BF16 ? 32 : 16);
  const bool hm = (g_msda_mode & 8) != 0;
#define MSDA_FWD(TT, HM_) msda_fwd_k<TT, HM_><<<blocks, 256, 0, ge_stream(stream)>>>((const TT*)value, lv, loc, attw, (TT*)out, n_groups, Nv, Nq, nH, L, P)
  if (dtype == GE_F32) { if (hm) MSDA_FWD(float, true); else MSDA_FWD(float, false); }
  else { if (hm) MSDA_FWD(bf16_t, true); else MSDA_FWD(bf16_t, false); }
#undef MSDA_FWD
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" size_t ge_msda_bwd_workspace(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P) {
  MsdaLevels lv;
  if (!spatial_hw || msda_levels(spatial_hw, L, Nv, lv)) return 0;
  MsdaBins bins;
  msda_bins(lv, L, bins);
  const MsdaPlan pl = msda_plan(bins, B, Nq, nH, L, P);
  if (!pl.ok) return 0;
  return msda_ws_layout(pl.nbins, (long)B * nH * pl.R * pl.ntiles, pl.max_entries, nullptr, nullptr);
}

// Which backward path ge_msda_bwd takes for a geometry (introspection for tests / DESIGN tables; no device work):
// out[0] = 1 binned (workspace) path available, out[1] = query ranges per (image, head) of the counting sort (its work units are
// (image, head, range); the LDS histogram of a unit holds one head's tiles), out[2] = value tiles, out[3] = bins.
extern "C" int ge_msda_bwd_plan(const int* spatial_hw, int B, int Nv, int Nq, int nH, int L, int P, int* out4) {
  MsdaLevels lv;
  if (!spatial_hw || !out4) return GE_ERR_BAD_ARG;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  MsdaBins bins;
  msda_bins(lv, L, bins);
  const MsdaPlan pl = msda_plan(bins, B, Nq, nH, L, P);
  out4[0] = pl.ok ? 1 : 0; out4[1] = pl.R; out4[2] = pl.ntiles; out4[3] = pl.nbins;
  return GE_OK;
}

// ---- optional per-kernel timing of the composite backward: a measurement aid for bench.py (its roofline object needs
// the duration of ONE kernel, and HIP events recorded by the caller can only bracket the whole entry point).  Off by
// default; when off the entry point records nothing and never synchronises.
#define MSDA_NSTAGE 5
static bool g_msda_lw_win_used = false;
static bool g_msda_drain_mfma_used = false;
static const char* const kMsdaStage[MSDA_NSTAGE] = {"msda_bwd_lw_k", "msda_hist_k<false>", "msda_scan_k+msda_segscan_k",
                                                    "msda_hist_k<true>", "msda_drain_k"};
struct MsdaStageRec { int stage; hipEvent_t a, b; };
static std::mutex g_msda_mu;
static bool g_msda_timing = false;
static std::vector<MsdaStageRec> g_msda_pending;
static double g_msda_ms[MSDA_NSTAGE];
static long g_msda_n[MSDA_NSTAGE];

static void msda_mark(hipEvent_t* ev, int i, hipStream_t s) {
  if (!ev) return;
  (void)hipEventCreate(&ev[i]);
  (void)hipEventRecord(ev[i], s);
}
static void msda_flush_locked() {
  for (auto& r : g_msda_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_msda_ms[r.stage] += ms;
      g_msda_n[r.stage] += 1;
    }
  }
  // every event is shared by two records (end of one stage, start of the next): destroy each once
  std::vector<hipEvent_t> seen;
  for (auto& r : g_msda_pending)
    for (hipEvent_t e : {r.a, r.b}) {
      bool dup = false;
      for (hipEvent_t x : seen) dup |= (x == e);
      if (!dup) { seen.push_back(e); (void)hipEventDestroy(e); }
    }
  g_msda_pending.clear();
}

extern "C" int ge_msda_bwd_timing(int enable) {
  std::lock_guard<std::mutex> lk(g_msda_mu);
  msda_flush_locked();
  for (int i = 0; i < MSDA_NSTAGE; ++i) { g_msda_ms[i] = 0.0; g_msda_n[i] = 0; }
  g_msda_timing = enable != 0;
  return GE_OK;
}
extern "C" int ge_msda_bwd_timing_read(int stage, double* total_ms, long* launches, char* name, int name_cap) {
  if (stage < 0 || stage >= MSDA_NSTAGE || !total_ms || !launches) return GE_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_msda_mu);
  msda_flush_locked();
  *total_ms = g_msda_ms[stage];
  *launches = g_msda_n[stage];
  const char* nm = (stage == 0 && g_msda_lw_win_used) ? "msda_bwd_lw_win_k" : (stage == 4 && g_msda_drain_mfma_used) ? "msda_drain_mfma_k" : kMsdaStage[stage];
  if (name && name_cap > 0) { strncpy(name, nm, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
  return GE_OK;
}

static int msda_bwd_impl(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc,
                         const float* attw, const void* d_out, float* d_value, float* d_loc, float* d_attw, void* workspace,
                         size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream, const MsdaEmit* em,
                         bool value_only = false) {
  if (!value || !spatial_hw || !loc || !attw || !d_out || (!d_value && !em) || (!em && !value_only && (!d_loc || !d_attw))) return GE_ERR_BAD_ARG;
  if (B < 0 || Nv <= 0 || Nq < 0 || nH <= 0 || P <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long n_groups = (long)B * Nq * nH;
  if (n_groups == 0) return GE_OK;
  const unsigned blocks = ge_blocks(n_groups, 4, 256 * 32);       // 4 waves per workgroup, persistent over groups
  hipStream_t s = ge_stream(stream);
  if (P != 4 && P != 8) return GE_ERR_UNSUPPORTED;
  if (dtype != GE_F32 && dtype != GE_BF16) return GE_ERR_UNSUPPORTED;
  // With a workspace: d_loc / d_attw by a read-only pass, d_value by bin -> register-tile accumulation (see above).
  // Without one: a single pass with coalesced 256-byte fp32 atomic bursts (bounded by the L2 atomic units).
#define MSDA_BWD(TT, PP, AT)                                                                                               \
  msda_bwd_k<TT, PP, AT><<<blocks, 256, 0, s>>>((const TT*)value, lv, loc, attw, (const TT*)d_out, d_value, d_loc, d_attw, \
                                                n_groups, Nv, Nq, nH, L)
#define MSDA_BWD_P(TT, AT) do { if (P == 8) MSDA_BWD(TT, 8, AT); else MSDA_BWD(TT, 4, AT); } while (0)
  MsdaBins bins;
  msda_bins(lv, L, bins);
  const MsdaPlan pl = msda_plan(bins, B, Nq, nH, L, P);
  const long seg_ints = (long)B * nH * pl.R * pl.ntiles;
  const bool binned = workspace && pl.ok && workspace_bytes >= msda_ws_layout(pl.nbins, seg_ints, pl.max_entries, nullptr, nullptr);
  if (!binned) {
    if (em || value_only) return GE_ERR_UNSUPPORTED;          // the raw-gradient / value-only variants exist on the workspace path only
    if (dtype == GE_F32) MSDA_BWD_P(float, true); else MSDA_BWD_P(bf16_t, true);
    GE_LAUNCH_CHECK();
    return GE_OK;
  }
#undef MSDA_BWD_P
#undef MSDA_BWD
  hipEvent_t evs[MSDA_NSTAGE + 1];
  hipEvent_t* ev = nullptr;
  { std::lock_guard<std::mutex> lk(g_msda_mu); if (g_msda_timing) ev = evs; }
  msda_mark(ev, 0, s);
  if (value_only) {
    // d_loc / d_attw come from ge_msda_bwd_lw_mm: only the binned d_value scatter runs here
  } else if (em) {
    if (L != 4 || P != 8) return GE_ERR_UNSUPPORTED;
    const unsigned lblocks = msda_grid(n_groups, dtype == GE_BF16 ? 32 : 16);
    const bool hm = (g_msda_mode & 8) != 0;
    g_msda_lw_win_used = false;
#define MSDA_LWE(TT, HM_) msda_bwd_lw_k<TT, HM_, true><<<lblocks, 256, 0, s>>>((const TT*)value, lv, loc, attw, (const TT*)d_out, nullptr, nullptr, n_groups, Nv, Nq, nH, L, P, *em)
    if (dtype == GE_F32) { if (hm) MSDA_LWE(float, true); else MSDA_LWE(float, false); }
    else { if (hm) MSDA_LWE(bf16_t, true); else MSDA_LWE(bf16_t, false); }
#undef MSDA_LWE
  } else if ((g_msda_mode & 2) && query_hw && n_qseg > 0 && msda_win_supported(B, Nq, nH, L, P, Nv)) {
    e = msda_bwd_lw_win_launch(value, lv, query_hw, n_qseg, loc, attw, d_out, d_loc, d_attw, B, Nv, Nq, nH, L, P, dtype, (g_msda_mode & 4) != 0, s);
    if (e) return e;
    g_msda_lw_win_used = true;
  } else {
    const unsigned lblocks = msda_grid(n_groups, dtype == GE_BF16 ? 32 : 16);   // one trip per workgroup, in query order
    const bool hm = (g_msda_mode & 8) != 0;
    g_msda_lw_win_used = false;
#define MSDA_LW(TT, HM_) msda_bwd_lw_k<TT, HM_><<<lblocks, 256, 0, s>>>((const TT*)value, lv, loc, attw, (const TT*)d_out, d_loc, d_attw, n_groups, Nv, Nq, nH, L, P)
    if (dtype == GE_F32) { if (hm) MSDA_LW(float, true); else MSDA_LW(float, false); }
    else { if (hm) MSDA_LW(bf16_t, true); else MSDA_LW(bf16_t, false); }
#undef MSDA_LW
  }
  GE_LAUNCH_CHECK();
  msda_mark(ev, 1, s);
  if (!d_value) {                                           // raw-gradient half only: the caller takes d_value from ge_msda_bwd_value_raw
    if (ev) { std::lock_guard<std::mutex> lk(g_msda_mu); g_msda_pending.push_back(MsdaStageRec{0, ev[0], ev[1]}); }
    return GE_OK;
  }
  const int nbins = pl.nbins;
  MsdaWs ws;
  msda_ws_layout(nbins, seg_ints, pl.max_entries, (char*)workspace, &ws);
  hipError_t he = hipMemsetAsync(ws.cnt, 0, (size_t)nbins * 4, s);
  if (he != hipSuccess) return (int)he;
  const unsigned hgrid = (unsigned)std::min((long)MSDA_HIST_WGS, (long)B * nH * pl.R);
  const size_t hsmem = (size_t)pl.ntiles * 4;
  msda_hist_k<false><<<hgrid, 1024, hsmem, s>>>(lv, bins, loc, attw, ws, Nq, nH, L, P, pl.R, B);
  GE_LAUNCH_CHECK();
  msda_mark(ev, 2, s);
  msda_scan_k<<<1, 1024, 0, s>>>(ws, nbins);
  GE_LAUNCH_CHECK();
  msda_segscan_k<<<(unsigned)((nbins + 255) / 256), 256, 0, s>>>(ws, pl.ntiles, pl.R, nbins);
  GE_LAUNCH_CHECK();
  if (g_msda_mode & 64) {
    msda_order_k<<<(unsigned)(B * nH), 1024, 0, s>>>(ws, pl.ntiles, pl.R, 1);
    GE_LAUNCH_CHECK();
  } else ws.order = nullptr;                               // bin order
  msda_mark(ev, 3, s);
  msda_hist_k<true><<<hgrid, 1024, hsmem, s>>>(lv, bins, loc, attw, ws, Nq, nH, L, P, pl.R, B);
  GE_LAUNCH_CHECK();
  msda_mark(ev, 4, s);
  g_msda_drain_mfma_used = dtype == GE_BF16 && (g_msda_mode & 16);
  if (dtype == GE_F32)
    msda_drain_k<float><<<256 * 3, 256, 0, s>>>(lv, bins, ws, (const float*)d_out, d_value, nbins, Nv, Nq, nH, L);
  else if (g_msda_drain_mfma_used) {
    e = msda_drain_mfma_launch(lv, bins, ws, d_out, d_value, nbins, Nv, Nq, nH, L, (g_msda_mode & 32) != 0, s);
    if (e) return e;
  } else
    msda_drain_k<bf16_t><<<256 * 3, 256, 0, s>>>(lv, bins, ws, (const bf16_t*)d_out, d_value, nbins, Nv, Nq, nH, L);
  GE_LAUNCH_CHECK();
  msda_mark(ev, 5, s);
  if (ev) {
    std::lock_guard<std::mutex> lk(g_msda_mu);
    for (int i = 0; i < MSDA_NSTAGE; ++i) g_msda_pending.push_back(MsdaStageRec{i, ev[i], ev[i + 1]});
  }
  return GE_OK;
}

extern "C" int ge_msda_bwd(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc,
                           const float* attw, const void* d_out, float* d_value, float* d_loc, float* d_attw, void* workspace,
                           size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  return msda_bwd_impl(value, spatial_hw, query_hw, n_qseg, loc, attw, d_out, d_value, d_loc, d_attw, workspace, workspace_bytes, B, Nv, Nq,
                       nH, L, P, dtype, stream, nullptr);
}

// d_value only (count / scan / fill / drain on the caller's workspace): the d_loc / d_attw half is ge_msda_bwd_lw_mm's
extern "C" int ge_msda_bwd_value(const void* value, const int* spatial_hw, const float* loc, const float* attw, const void* d_out,
                                 float* d_value, void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P,
                                 int dtype, void* stream) {
  return msda_bwd_impl(value, spatial_hw, nullptr, 0, loc, attw, d_out, d_value, nullptr, nullptr, workspace, workspace_bytes, B, Nv, Nq,
                       nH, L, P, dtype, stream, nullptr, true);
}

// d_value from the raw projections (bf16 storage, L == 4, P == 8): count / scan / fill with 8-byte records, MFMA drain.  Same workspace
// as ge_msda_bwd (ge_msda_bwd_workspace); d_value f32, zero-filled by the caller.
extern "C" int ge_msda_bwd_value_raw(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                                     const float* ref, long ref_sb, long ref_sq, long ref_sl, const void* d_out, float* d_value,
                                     void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype,
                                     void* stream) {
  return ge_msda_bwd_value_raw_levels(spatial_hw, off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, d_out, d_value, workspace,
                                      workspace_bytes, 15, B, Nv, Nq, nH, L, P, dtype, stream);
}

// The same restricted to the levels of `level_mask` (bit l): the other levels' rows of d_value are left untouched — they come from
// ge_msda_bwd_value_mm, whose cost depends on the level (coarse levels: long runs of query tiles share a window).
extern "C" int ge_msda_bwd_value_raw_levels(const int* spatial_hw, const void* off_raw, long off_ld, const void* logit_raw, long logit_ld,
                                            const float* ref, long ref_sb, long ref_sq, long ref_sl, const void* d_out, float* d_value,
                                            void* workspace, size_t workspace_bytes, int level_mask, int B, int Nv, int Nq, int nH, int L,
                                            int P, int dtype, void* stream) {
  if (!spatial_hw || !off_raw || !logit_raw || !ref || !d_out || !d_value || !workspace || B < 0 || Nq < 0 || nH <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_BF16 || L != 4 || P != 8) return GE_ERR_UNSUPPORTED;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  for (int l = 0; l < L; ++l) if (lv.W[l] > 8191 || lv.H[l] > 8191) return GE_ERR_UNSUPPORTED;
  if ((long)B * Nq * nH == 0) return GE_OK;
  MsdaBins bins;
  msda_bins(lv, L, bins);
  const MsdaPlan pl = msda_plan(bins, B, Nq, nH, L, P);
  const long seg_ints = (long)B * nH * pl.R * pl.ntiles;
  if (!pl.ok || workspace_bytes < msda_ws_layout(pl.nbins, seg_ints, pl.max_entries, nullptr, nullptr)) return GE_ERR_UNSUPPORTED;
  hipStream_t s = ge_stream(stream);
  MsdaRawIn in;
  in.off = (const bf16_t*)off_raw; in.off_ld = off_ld; in.logit = (const bf16_t*)logit_raw; in.logit_ld = logit_ld;
  in.ref = ref; in.ref_sb = ref_sb; in.ref_sq = ref_sq; in.ref_sl = ref_sl;
  for (int l = 0; l < 4; ++l) { in.fW[l] = (float)lv.W[l]; in.fH[l] = (float)lv.H[l]; in.rW[l] = 1.f / in.fW[l]; in.rH[l] = 1.f / in.fH[l]; }
  hipEvent_t evs[MSDA_NSTAGE + 1];
  hipEvent_t* ev = nullptr;
  { std::lock_guard<std::mutex> lk(g_msda_mu); if (g_msda_timing) ev = evs; }
  msda_mark(ev, 0, s);
  msda_mark(ev, 1, s);
  const int nbins = pl.nbins;
  MsdaWs ws;
  msda_ws_layout(nbins, seg_ints, pl.max_entries, (char*)workspace, &ws);
  hipError_t he = hipMemsetAsync(ws.cnt, 0, (size_t)nbins * 4, s);
  if (he != hipSuccess) return (int)he;
  const unsigned hgrid = (unsigned)std::min((long)MSDA_HIST_WGS, (long)B * nH * pl.R);
  const size_t hsmem = (size_t)pl.ntiles * 4;
  msda_hist_raw_k<false><<<hgrid, 1024, hsmem, s>>>(lv, bins, in, ws, Nq, nH, pl.R, B, level_mask);
  GE_LAUNCH_CHECK();
  msda_mark(ev, 2, s);
  msda_scan_k<<<1, 1024, 0, s>>>(ws, nbins);
  GE_LAUNCH_CHECK();
  msda_segscan_k<<<(unsigned)((nbins + 255) / 256), 256, 0, s>>>(ws, pl.ntiles, pl.R, nbins);
  GE_LAUNCH_CHECK();
  if (g_msda_mode & 64) {
    msda_order_k<<<(unsigned)(B * nH), 1024, 0, s>>>(ws, pl.ntiles, pl.R, 1);
    GE_LAUNCH_CHECK();
  } else ws.order = nullptr;                               // bin order
  msda_mark(ev, 3, s);
  msda_hist_raw_k<true><<<hgrid, 1024, hsmem, s>>>(lv, bins, in, ws, Nq, nH, pl.R, B, level_mask);
  GE_LAUNCH_CHECK();
  msda_mark(ev, 4, s);
  g_msda_drain_mfma_used = true;
  g_msda_lw_win_used = false;
  e = msda_drain_mfma_launch(lv, bins, ws, d_out, d_value, nbins, Nv, Nq, nH, L, true, s, true);
  if (e) return e;
  msda_mark(ev, 5, s);
  if (ev) {
    std::lock_guard<std::mutex> lk(g_msda_mu);
    for (int i = 0; i < MSDA_NSTAGE; ++i) g_msda_pending.push_back(MsdaStageRec{i, ev[i], ev[i + 1]});
  }
  return GE_OK;
}

// ============================================================================ sampling-location / weight preparation
// One pass from the raw outputs of the `sampling_offsets` and `attention_weights` linears to the fp32 tensors the
// sampling kernels read (mmcv MultiScaleDeformableAttention.forward: view -> softmax over L*P -> offset_normalizer
// -> reference_points + offsets / normalizer):
//   loc [b,q,h,l,p,:] = ref[b,q,l,:] + off_raw[b,q,(h,l,p,:)] / (W_l, H_l)
//   attw[b,q,h,l,p]   = softmax over (l,p) of logit_raw[b,q,(h,l,p)]
// One thread per (b,q,h,l) handles its P points; the softmax statistics cross the L = 4 neighbouring lanes with
// DPP-style shuffles.  Replaces two dtype casts, a divide, an add and a softmax (each a full pass over 0.8-1.6 GB at
// the KITTI shape) and their five backward passes.
template <typename T, int K>
__device__ __forceinline__ void ld_run(const T* p, float* v) {       // K contiguous elements, 16-byte vectors when K allows
  constexpr int VN = V8<T>::N;
  if constexpr (K % VN == 0) {
#pragma unroll
    for (int i = 0; i < K; i += VN) V8<T>::ld(p + i, v + i);
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = Io<T>::ld(p + i);
  }
}
template <typename T, int K>
__device__ __forceinline__ void st_run(T* p, const float* v) {
  constexpr int VN = V8<T>::N;
  if constexpr (K % VN == 0) {
#pragma unroll
    for (int i = 0; i < K; i += VN) V8<T>::st(p + i, v + i);
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) Io<T>::st(p + i, v[i]);
  }
}

template <typename T, int P>
__global__ void __launch_bounds__(256) msda_prep_fwd_k(const T* __restrict__ off_raw, long off_ld, const T* __restrict__ logit_raw,
                                                       long logit_ld, const float* __restrict__ ref, long ref_sb, long ref_sq,
                                                       long ref_sl, MsdaLevels lv, float* __restrict__ loc, float* __restrict__ attw,
                                                       long total, int Nq, int nH) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int l = (int)(t & 3);
    const long g = t >> 2;
    const int h = (int)(g % nH);
    const long row = g / nH;
    const long b = row / Nq, q = row - b * Nq;
    float off[2 * P], lg[P];
    ld_run<T, 2 * P>(off_raw + row * off_ld + ((long)h * 4 + l) * (2 * P), off);
    ld_run<T, P>(logit_raw + row * logit_ld + ((long)h * 4 + l) * P, lg);
    const float* rp = ref + b * ref_sb + q * ref_sq + l * ref_sl;
    const float rx = rp[0], ry = rp[1];
    const float W = (float)lv.W[l], H = (float)lv.H[l];
    float m = lg[0];
#pragma unroll
    for (int i = 1; i < P; ++i) m = fmaxf(m, lg[i]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) { lg[i] = expf(lg[i] - m); s += lg[i]; }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    float o[2 * P];
#pragma unroll
    for (int i = 0; i < P; ++i) { o[2 * i] = rx + off[2 * i] / W; o[2 * i + 1] = ry + off[2 * i + 1] / H; lg[i] = lg[i] / s; }
    st_run<float, 2 * P>(loc + t * (2 * P), o);
    st_run<float, P>(attw + t * P, lg);
  }
}

// d_off_raw = d_loc / (W_l, H_l);  d_logit = attw * (d_attw - sum_{l,p} attw * d_attw);  d_ref[b,q,l,:] = sum_{h,p} d_loc
template <typename T, int P>
__global__ void __launch_bounds__(256) msda_prep_bwd_k(const float* __restrict__ d_loc, const float* __restrict__ d_attw,
                                                       const float* __restrict__ attw, MsdaLevels lv, T* __restrict__ d_off_raw,
                                                       long off_ld, T* __restrict__ d_logit_raw, long logit_ld,
                                                       float* __restrict__ d_ref, long total, int Nq, int nH) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int l = (int)(t & 3);
    const long g = t >> 2;
    const int h = (int)(g % nH);
    const long row = g / nH;
    float dl[2 * P], da[P], a[P];
    ld_run<float, 2 * P>(d_loc + t * (2 * P), dl);
    ld_run<float, P>(d_attw + t * P, da);
    ld_run<float, P>(attw + t * P, a);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) s += a[i] * da[i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
#pragma unroll
    for (int i = 0; i < P; ++i) da[i] = a[i] * (da[i] - s);
    if (d_ref) {                                   // nH * 4 lanes of one query are adjacent (launcher checks nH in {1,2,4,8,16})
      float sx = 0.f, sy = 0.f;
#pragma unroll
      for (int i = 0; i < P; ++i) { sx += dl[2 * i]; sy += dl[2 * i + 1]; }
      for (int o = 4; o < 4 * nH; o <<= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
      if (h == 0) { d_ref[(row * 4 + l) * 2] = sx; d_ref[(row * 4 + l) * 2 + 1] = sy; }
    }
    const float iW = (float)lv.W[l], iH = (float)lv.H[l];
#pragma unroll
    for (int i = 0; i < P; ++i) { dl[2 * i] = dl[2 * i] / iW; dl[2 * i + 1] = dl[2 * i + 1] / iH; }
    st_run<T, 2 * P>(d_off_raw + row * off_ld + ((long)h * 4 + l) * (2 * P), dl);
    st_run<T, P>(d_logit_raw + row * logit_ld + ((long)h * 4 + l) * P, da);
  }
}

static int prep_levels(const int* spatial_hw, int L, MsdaLevels& lv) {
  if (L != 4) return GE_ERR_UNSUPPORTED;                                  // the 4-lane softmax exchange is the level axis
  for (int l = 0; l < L; ++l) {
    lv.H[l] = spatial_hw[2 * l]; lv.W[l] = spatial_hw[2 * l + 1]; lv.start[l] = 0;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return GE_ERR_BAD_ARG;
  }
  return GE_OK;
}
static bool prep_aligned(const void* a, long a_ld, const void* b, long b_ld, int esize) {
  const long v = 16 / esize;
  return ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0 && a_ld % v == 0 && b_ld % v == 0;
}

extern "C" int ge_msda_prep_fwd(const void* off_raw, long off_ld, const void* logit_raw, long logit_ld, const float* ref,
                                long ref_sb, long ref_sq, long ref_sl, const int* spatial_hw, float* loc, float* attw, int B,
                                int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!off_raw || !logit_raw || !ref || !spatial_hw || !loc || !attw || B < 0 || Nq < 0 || nH <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = prep_levels(spatial_hw, L, lv);
  if (e) return e;
  if (P != 4 && P != 8) return GE_ERR_UNSUPPORTED;
  if (!prep_aligned(off_raw, off_ld, logit_raw, logit_ld, dtype == GE_BF16 ? 2 : 4)) return GE_ERR_BAD_ARG;
  const long total = (long)B * Nq * nH * 4;
  if (total == 0) return GE_OK;
  const unsigned blocks = ge_blocks(total, 256, 1 << 20);
  hipStream_t s = ge_stream(stream);
#define GE_PREP(T, PP) msda_prep_fwd_k<T, PP><<<blocks, 256, 0, s>>>((const T*)off_raw, off_ld, (const T*)logit_raw, logit_ld, ref, \
                                                                      ref_sb, ref_sq, ref_sl, lv, loc, attw, total, Nq, nH)
  if (dtype == GE_F32) { if (P == 8) GE_PREP(float, 8); else GE_PREP(float, 4); }
  else if (dtype == GE_BF16) { if (P == 8) GE_PREP(bf16_t, 8); else GE_PREP(bf16_t, 4); }
  else return GE_ERR_UNSUPPORTED;
#undef GE_PREP
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_msda_prep_bwd(const float* d_loc, const float* d_attw, const float* attw, const int* spatial_hw, void* d_off_raw,
                                long off_ld, void* d_logit_raw, long logit_ld, float* d_ref, int B, int Nq, int nH, int L, int P,
                                int dtype, void* stream) {
  if (!d_loc || !d_attw || !attw || !spatial_hw || !d_off_raw || !d_logit_raw || B < 0 || Nq < 0 || nH <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = prep_levels(spatial_hw, L, lv);
  if (e) return e;
  if (P != 4 && P != 8) return GE_ERR_UNSUPPORTED;
  if (d_ref && !(nH == 1 || nH == 2 || nH == 4 || nH == 8 || nH == 16)) return GE_ERR_UNSUPPORTED;
  if (!prep_aligned(d_off_raw, off_ld, d_logit_raw, logit_ld, dtype == GE_BF16 ? 2 : 4)) return GE_ERR_BAD_ARG;
  const long total = (long)B * Nq * nH * 4;
  if (total == 0) return GE_OK;
  const unsigned blocks = ge_blocks(total, 256, 1 << 20);
  hipStream_t s = ge_stream(stream);
#define GE_PREP(T, PP) msda_prep_bwd_k<T, PP><<<blocks, 256, 0, s>>>(d_loc, d_attw, attw, lv, (T*)d_off_raw, off_ld, (T*)d_logit_raw, \
                                                                      logit_ld, d_ref, total, Nq, nH)
  if (dtype == GE_F32) { if (P == 8) GE_PREP(float, 8); else GE_PREP(float, 4); }
  else if (dtype == GE_BF16) { if (P == 8) GE_PREP(bf16_t, 8); else GE_PREP(bf16_t, 4); }
  else return GE_ERR_UNSUPPORTED;
#undef GE_PREP
  GE_LAUNCH_CHECK();
  return GE_OK;
}


// ============================================================================ fused prepare + sampling ("raw" entry points)
// d_ref[b, q, l, :] = sum_{h, p} d_loc[b, q, h, l, p, :] from the emitted d_off_raw = d_loc / (W_l, H_l) (the heads of a query are
// spread over workgroups in the head-major d_loc / d_attw kernel, so the sum is a small pass of its own: one thread per (row, level)).
// Parity note: with bf16 storage d_off_raw is already rounded to bf16 when it is read back here, so d_ref carries one bf16 rounding per
// (head, point) term (2^-9 relative each, nH * P = 64 terms per reference point and level) that the composed path (ge_msda_prep_bwd, which
// sums the fp32 d_loc) does not have; fp32 storage is exact either way.  The consumer is the gradient of the `reference_points` Linear of
// the cross-attention (hahi.py:294-302), whose other operand is bf16 as well; tests/test_kernels_gpu.py bounds it at the bf16 tolerance.
template <typename T>
__global__ void __launch_bounds__(256) msda_dref_k(const T* __restrict__ d_off, long off_ld, MsdaLevels lv, float* __restrict__ d_ref, long rows,
                                                   int nH) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < rows * 4; t += (long)gridDim.x * 256) {
    const int l = (int)(t & 3);
    const long row = t >> 2;
    float sx = 0.f, sy = 0.f;
    for (int h = 0; h < nH; ++h) {
      float v[16];
      ld_run<T, 16>(d_off + row * off_ld + ((long)h * 4 + l) * 16, v);
#pragma unroll
      for (int p = 0; p < 8; ++p) { sx += v[2 * p]; sy += v[2 * p + 1]; }
    }
    d_ref[t * 2] = sx * (float)lv.W[l];
    d_ref[t * 2 + 1] = sy * (float)lv.H[l];
  }
}


// 1 when ge_msda_fwd_raw / ge_msda_bwd_raw run their fused kernels for this geometry under the current kernel-selection mode
extern "C" int ge_msda_raw_supported(const int* spatial_hw, const int* query_hw, int n_qseg, int B, int Nv, int Nq, int nH, int L, int P) {
  MsdaLevels lv;
  if (!spatial_hw || msda_levels(spatial_hw, L, Nv, lv)) return 0;
  if (L != 4 || P != 8 || !query_hw || n_qseg <= 0 || (g_msda_mode & 5) != 5 || (g_msda_mode & 2)) return 0;
  if (!msda_win_supported(B, Nq, nH, L, P, Nv)) return 0;
  MsdaBins bins;
  msda_bins(lv, L, bins);
  return msda_plan(bins, B, Nq, nH, L, P).ok ? 1 : 0;
}

// Forward from the raw projection outputs: off_raw (B*Nq rows, columns (head, level, point, xy)), logit_raw (columns (head, level,
// point)), reference points (B, Nq, L, 2) with element strides -> out, and loc / attw (fp32, fully written) for the backward.
extern "C" int ge_msda_fwd_raw(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const void* off_raw, long off_ld,
                               const void* logit_raw, long logit_ld, const float* ref, long ref_sb, long ref_sq, long ref_sl, float* loc,
                               float* attw, void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !off_raw || !logit_raw || !ref || !loc || !attw || !out) return GE_ERR_BAD_ARG;
  if (dtype != GE_F32 && dtype != GE_BF16) return GE_ERR_UNSUPPORTED;
  if (!ge_msda_raw_supported(spatial_hw, query_hw, n_qseg, B, Nv, Nq, nH, L, P)) {       // composed: prepare pass, then the sampling kernel
    int e = ge_msda_prep_fwd(off_raw, off_ld, logit_raw, logit_ld, ref, ref_sb, ref_sq, ref_sl, spatial_hw, loc, attw, B, Nq, nH, L, P, dtype, stream);
    if (e) return e;
    return ge_msda_fwd(value, spatial_hw, query_hw, n_qseg, loc, attw, out, B, Nv, Nq, nH, L, P, dtype, stream);
  }
  if ((long)B * Nq == 0) return GE_OK;
  const int es = dtype == GE_BF16 ? 2 : 4;
  if ((((uintptr_t)off_raw | (uintptr_t)logit_raw) & 15) || off_ld % (16 / es) || logit_ld % (16 / es)) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  MwRaw rw;
  rw.off = off_raw; rw.off_ld = off_ld; rw.logit = logit_raw; rw.logit_ld = logit_ld;
  rw.ref = ref; rw.ref_sb = ref_sb; rw.ref_sq = ref_sq; rw.ref_sl = ref_sl; rw.loc_out = loc; rw.attw_out = attw;
  return msda_fwd_win_raw_launch(value, lv, query_hw, n_qseg, rw, out, B, Nv, Nq, nH, L, P, dtype, ge_stream(stream));
}

// Backward to the raw projection outputs: d_value (fp32, zero-filled by the caller; NULL = skip the d_value scatter, e.g. when the
// caller takes it from ge_msda_bwd_value_raw), d_off_raw / d_logit_raw (storage type, fully written) and, if d_ref != NULL, d_ref
// (B*Nq, L, 2) fp32.  Needs the workspace of ge_msda_bwd_workspace and a geometry for which
// ge_msda_raw_supported() is 1.
extern "C" int ge_msda_bwd_raw(const void* value, const int* spatial_hw, const int* query_hw, int n_qseg, const float* loc, const float* attw,
                               const void* d_out, float* d_value, void* d_off_raw, long off_ld, void* d_logit_raw, long logit_ld, float* d_ref,
                               void* workspace, size_t workspace_bytes, int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!d_off_raw || !d_logit_raw) return GE_ERR_BAD_ARG;
  if (!ge_msda_raw_supported(spatial_hw, query_hw, n_qseg, B, Nv, Nq, nH, L, P)) return GE_ERR_UNSUPPORTED;
  if (!prep_aligned(d_off_raw, off_ld, d_logit_raw, logit_ld, dtype == GE_BF16 ? 2 : 4)) return GE_ERR_BAD_ARG;
  MsdaEmit em;
  em.d_off = d_off_raw; em.off_ld = off_ld; em.d_logit = d_logit_raw; em.logit_ld = logit_ld;
  int e = msda_bwd_impl(value, spatial_hw, query_hw, n_qseg, loc, attw, d_out, d_value, nullptr, nullptr, workspace, workspace_bytes, B, Nv, Nq,
                        nH, L, P, dtype, stream, &em);
  if (e || !d_ref) return e;
  MsdaLevels lv;
  e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long rows = (long)B * Nq;
  if (rows == 0) return GE_OK;
  const unsigned blocks = ge_blocks(rows * 4, 256, 1 << 20);
  if (dtype == GE_F32) msda_dref_k<float><<<blocks, 256, 0, ge_stream(stream)>>>((const float*)d_off_raw, off_ld, lv, d_ref, rows, nH);
  else msda_dref_k<bf16_t><<<blocks, 256, 0, ge_stream(stream)>>>((const bf16_t*)d_off_raw, off_ld, lv, d_ref, rows, nH);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// d_ref (B*Nq, L, 2) f32 = sum over heads and points of d_loc, rebuilt from the emitted d_off_raw = d_loc / (W_l, H_l) (L == 4, P == 8)
extern "C" int ge_msda_dref(const void* d_off_raw, long off_ld, const int* spatial_hw, float* d_ref, long rows, int nH, int L, int P, int dtype,
                            void* stream) {
  if (!d_off_raw || !spatial_hw || !d_ref || rows < 0 || nH <= 0) return GE_ERR_BAD_ARG;
  if (L != 4 || P != 8 || (dtype != GE_F32 && dtype != GE_BF16)) return GE_ERR_UNSUPPORTED;
  MsdaLevels lv;
  int e = prep_levels(spatial_hw, L, lv);
  if (e) return e;
  if (rows == 0) return GE_OK;
  const unsigned blocks = ge_blocks(rows * 4, 256, 1 << 20);
  if (dtype == GE_F32) msda_dref_k<float><<<blocks, 256, 0, ge_stream(stream)>>>((const float*)d_off_raw, off_ld, lv, d_ref, rows, nH);
  else msda_dref_k<bf16_t><<<blocks, 256, 0, ge_stream(stream)>>>((const bf16_t*)d_off_raw, off_ld, lv, d_ref, rows, nH);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
