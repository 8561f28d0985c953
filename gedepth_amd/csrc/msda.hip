// Multi-scale deformable attention sampling core for gfx950 (replaces mmcv's ms_deform_attn CUDA op,
// reference call sites depth/models/necks/hahi.py:279-289,316-325).
//
// Work decomposition (wave64-first): a group of 16 lanes owns one (batch, query, head); each lane owns
// 4 of the head's 64 channels, so every bilinear tap is one 256-byte (fp32) / 128-byte (bf16)
// fully-coalesced read of `value`.  A 256-thread workgroup therefore covers 16 (query, head) pairs =
// two whole queries, whose sampling locations / attention weights are contiguous in memory.
// The op is a gather (no contraction) -> no MFMA; it is bound by L2/MALL gather bandwidth.
// Backward accumulates d_value with fp32 hardware atomics (global_atomic_add_f32) and reduces
// d_loc / d_attw over the 64 channels with 16-lane butterfly shuffles.
#include <algorithm>
#include <cstdlib>
#include "common.h"

#define MSDA_MAX_L 8
struct MsdaLevels { int H[MSDA_MAX_L]; int W[MSDA_MAX_L]; int start[MSDA_MAX_L]; };

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

template <typename T>
__global__ void __launch_bounds__(256) msda_fwd_k(const T* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                  const float* __restrict__ attw, T* __restrict__ out,
                                                  long n_groups, int Nv, int Nq, int nH, int L, int P) {
  const int c4 = (threadIdx.x & 15) * 4;
  const long grp0 = (long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  const long gstride = (long)gridDim.x * (blockDim.x >> 4);
  for (long grp = grp0; grp < n_groups; grp += gstride) {   // grp = (b*Nq + q)*nH + head
    const int head = (int)(grp % nH);
    const long bq = grp / nH;
    const int b = (int)(bq / Nq);
    const float* lp = loc + grp * (long)(L * P * 2);
    const float* ap = attw + grp * (long)(L * P);
    const T* vb = value + ((long)b * Nv * nH + head) * 64 + c4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const T* vl = vb + (long)lv.start[l] * nH * 64;
      for (int p = 0; p < P; ++p) {
        const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
        const float wgt = ap[l * P + p];
        const float x = lx * (float)Wl - 0.5f, y = ly * (float)Hl - 0.5f;   // grid_sample, align_corners=False
        if (!(y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl)) continue;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
        const float ax = x - xf, ay = y - yf;
        const float w00 = (1.f - ay) * (1.f - ax) * wgt, w01 = (1.f - ay) * ax * wgt;
        const float w10 = ay * (1.f - ax) * wgt, w11 = ay * ax * wgt;
        float v[4];
        if (y0 >= 0 && x0 >= 0) { Vec4<T>::ld(vl + ((long)y0 * Wl + x0) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w00 * v[i]; }
        if (y0 >= 0 && x1 < Wl) { Vec4<T>::ld(vl + ((long)y0 * Wl + x1) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w01 * v[i]; }
        if (y1 < Hl && x0 >= 0) { Vec4<T>::ld(vl + ((long)y1 * Wl + x0) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w10 * v[i]; }
        if (y1 < Hl && x1 < Wl) { Vec4<T>::ld(vl + ((long)y1 * Wl + x1) * nH * 64, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += w11 * v[i]; }
      }
    }
    Vec4<T>::st(out + grp * 64 + c4, acc);
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
    for (int c0 = 0; c0 < LP; c0 += 32) {                         // 32 points per chunk: 64 loc floats, 32 weights
      const int npt = min(32, LP - c0);
      const float locv = (lane < 2 * npt) ? lp[c0 * 2 + lane] : 0.f;
      const float attv = (lane < npt) ? ap[c0 + lane] : 0.f;
      float my_dattw = 0.f, my_dloc = 0.f;                        // results for point `lane` / loc float `lane`
      for (int j = 0; j < npt; ++j) {
        const int pt = c0 + j;
        const int l = pt / P;
        const int Hl = lv.H[l], Wl = lv.W[l];
        const float lx = readlane_f(locv, 2 * j), ly = readlane_f(locv, 2 * j + 1);
        const float wgt = readlane_f(attv, j);
        const float x = lx * (float)Wl - 0.5f, y = ly * (float)Hl - 0.5f;
        float s_val = 0.f, s_dx = 0.f, s_dy = 0.f;
        if (y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl) {          // wave-uniform branch
          const float xf = floorf(x), yf = floorf(y);
          const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
          const float ax = x - xf, ay = y - yf, bx = 1.f - ax, by = 1.f - ay;
          const long lbase = vbase + (long)lv.start[l] * nH * 64;
          const float gw = go * wgt;
          if (y0 >= 0 && x0 >= 0) {
            const long o = lbase + ((long)y0 * Wl + x0) * nH * 64;
            const float gv = go * Ld1<T>::ld(value + o);
            s_val += by * bx * gv; s_dx -= by * gv; s_dy -= bx * gv;
            if (VALUE_ATOMICS) atomicAdd(d_value + o, gw * (by * bx));
          }
          if (y0 >= 0 && x1 < Wl) {
            const long o = lbase + ((long)y0 * Wl + x1) * nH * 64;
            const float gv = go * Ld1<T>::ld(value + o);
            s_val += by * ax * gv; s_dx += by * gv; s_dy -= ax * gv;
            if (VALUE_ATOMICS) atomicAdd(d_value + o, gw * (by * ax));
          }
          if (y1 < Hl && x0 >= 0) {
            const long o = lbase + ((long)y1 * Wl + x0) * nH * 64;
            const float gv = go * Ld1<T>::ld(value + o);
            s_val += ay * bx * gv; s_dx -= ay * gv; s_dy += bx * gv;
            if (VALUE_ATOMICS) atomicAdd(d_value + o, gw * (ay * bx));
          }
          if (y1 < Hl && x1 < Wl) {
            const long o = lbase + ((long)y1 * Wl + x1) * nH * 64;
            const float gv = go * Ld1<T>::ld(value + o);
            s_val += ay * ax * gv; s_dx += ay * gv; s_dy += ax * gv;
            if (VALUE_ATOMICS) atomicAdd(d_value + o, gw * (ay * ax));
          }
        }
        s_val = wave_sum(s_val);
        s_dx = wave_sum(s_dx) * wgt * (float)Wl;
        s_dy = wave_sum(s_dy) * wgt * (float)Hl;
        if (lane == j) my_dattw = s_val;
        if (lane == 2 * j) my_dloc = s_dx;
        if (lane == 2 * j + 1) my_dloc = s_dy;
      }
      if (lane < npt) d_attw[grp * (long)LP + c0 + lane] = my_dattw;           // coalesced result rows
      if (lane < 2 * npt) d_loc[grp * (long)(LP * 2) + c0 * 2 + lane] = my_dloc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// d_value without per-tap atomics: "tile owner" scatter into REGISTERS.
// Measured on MI355X (scratch/ubench): the L2 executes fp32 atomics at ~1 dword/clock/channel (5.0 G 256-byte
// bursts/s chip-wide, independent of footprint and scope) and LDS fp32 atomics are slower still (ds_add_f32: ~170
// cycles per wave-instruction per CU), which pinned the plain scatter at ~140 ms for the cross-attention of 8 images.
// Here a WAVE owns a tile of 128 value positions of one (batch, head, level): lane == channel, and the tile lives in
// 128 accumulator VGPRs per lane.  The wave scans every sampling point of its level (one point per lane, coalesced
// loc / attw reads that stay L2-resident because all owners of a (batch, head) stream them together), queues the taps
// that land in its tile (ballot-compacted into a small LDS queue), and drains the queue wave-wide: one coalesced read
// of the query's gradient row per tap (16 in flight), then `acc[position] += g * coef` with the position as a
// wave-uniform dynamic register index (s_set_gpr_idx — no LDS, no atomics).  Coarse levels have few tiles but receive
// as many taps as the fine ones, so their query range is split over several waves (balanced tap count per wave) and
// only those partial tiles meet through atomics in the (zero-filled) output: ~0.1 % of the original atomic traffic.
#define MSDA_TILE 128
#define MSDA_QCAP 384          // per-wave tap queue (drained when fewer than 256 free slots remain)
#define MSDA_DRAIN_U 16        // gradient-row loads in flight per wave while draining
typedef float f32x32_t __attribute__((ext_vector_type(32)));
struct MsdaTileMap { int first_block[MSDA_MAX_L + 1]; int splits[MSDA_MAX_L]; };   // owners of level l: tiles_l x splits_l

template <typename T>
__global__ void __launch_bounds__(256) msda_bwd_value_k(MsdaLevels lv, MsdaTileMap tm, const float* __restrict__ loc,
                                                        const float* __restrict__ attw, const T* __restrict__ gout,
                                                        float* __restrict__ d_value, int Nv, int Nq, int nH, int L, int P,
                                                        int n_owner) {
  __shared__ int q_key[4 * MSDA_QCAP];        // query << 8 | position in tile
  __shared__ float q_coef[4 * MSDA_QCAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int owner = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);     // wave-uniform
  if (owner >= n_owner) return;
  const int bh = blockIdx.y;
  const int b = bh / nH, head = bh - b * nH;
  int l = 0;
  while (l + 1 < L && owner >= tm.first_block[l + 1]) ++l;
  const int Hl = lv.H[l], Wl = lv.W[l];
  const int nsplit = tm.splits[l];
  const int bidx = owner - tm.first_block[l];
  const int tile_lo = (bidx / nsplit) * MSDA_TILE;
  const int split = bidx % nsplit;
  const int tile_n = min(MSDA_TILE, Hl * Wl - tile_lo);
  const int q_lo = (int)((long)Nq * split / nsplit), q_hi = (int)((long)Nq * (split + 1) / nsplit);

  f32x32_t a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int* mykey = q_key + wave * MSDA_QCAP;
  float* mycoef = q_coef + wave * MSDA_QCAP;
  int qcount = 0;                                                        // wave-uniform
  const long rowbase = (long)b * Nq * nH + head;
  const int LP = L * P;
  const long total = (long)q_hi * P, first = (long)q_lo * P;
  const float fW = (float)Wl, fH = (float)Hl;

  // (macro, not a lambda: capturing the accumulator vectors by reference would force them into scratch memory)
#define MSDA_DRAIN()                                                                                        \
  do {                                                                                                      \
    for (int e = 0; e < qcount; e += MSDA_DRAIN_U) {                                                        \
      float g[MSDA_DRAIN_U];                                                                                \
      int key[MSDA_DRAIN_U];                                                                                \
      _Pragma("unroll") for (int k = 0; k < MSDA_DRAIN_U; ++k) {                                            \
        const int ee = min(e + k, qcount - 1);                                                              \
        key[k] = mykey[ee];                                                                                 \
        const float cfk = (e + k < qcount) ? mycoef[ee] : 0.f;                                              \
        g[k] = cfk * Ld1<T>::ld(gout + (rowbase + (long)(key[k] >> 8) * nH) * 64 + lane);                   \
      }                                                                                                     \
      _Pragma("unroll") for (int k = 0; k < MSDA_DRAIN_U; ++k) {                                            \
        const int r = __builtin_amdgcn_readfirstlane(key[k]) & 0xff;                                        \
        const int gsel = r >> 5, e5 = r & 31;                                                               \
        if (gsel == 0) a0[e5] += g[k];                                                                      \
        else if (gsel == 1) a1[e5] += g[k];                                                                 \
        else if (gsel == 2) a2[e5] += g[k];                                                                 \
        else a3[e5] += g[k];                                                                                \
      }                                                                                                     \
    }                                                                                                       \
    qcount = 0;                                                                                             \
  } while (0)
  auto fetch = [&](long i, float& lx, float& ly, float& wgt) {
    lx = 0.f; ly = -4.f; wgt = 0.f;                                      // rejected by the bounds test below
    if (i < total) {
      const int q = (int)(i / P);
      const int p = (int)(i - (long)q * P);
      const long row = rowbase + (long)q * nH;
      const float2 xy = *(const float2*)(loc + row * (LP * 2) + (l * P + p) * 2);
      lx = xy.x; ly = xy.y;
      wgt = attw[row * LP + l * P + p];
    }
  };
  float lx, ly, wgt;
  fetch(first + lane, lx, ly, wgt);
  for (long base = first; base < total; base += 64) {
    float nlx, nly, nwgt;
    fetch(base + 64 + lane, nlx, nly, nwgt);
    const int q = (int)((base + lane) / P);
    bool m[4] = {false, false, false, false};
    int rel[4] = {0, 0, 0, 0};
    float cf[4] = {0.f, 0.f, 0.f, 0.f};
    const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
    if (y > -1.f && x > -1.f && y < fH && x < fW) {
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float ax = x - xf, ay = y - yf, bx = 1.f - ax, by = 1.f - ay;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
        if (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) {
          const int r = yy * Wl + xx - tile_lo;
          if ((unsigned)r < (unsigned)tile_n) {
            m[t] = true;
            rel[t] = r;
            cf[t] = wgt * (((t >> 1) ? ay : by) * ((t & 1) ? ax : bx));
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned long long mask = __ballot(m[t]);
      if (m[t]) {
        const int slot = qcount + __popcll(mask & ((1ull << lane) - 1ull));
        mykey[slot] = (q << 8) | rel[t];
        mycoef[slot] = cf[t];
      }
      qcount += __popcll(mask);
    }
    if (qcount > MSDA_QCAP - 256) MSDA_DRAIN();
    lx = nlx; ly = nly; wgt = nwgt;
  }
  MSDA_DRAIN();
  float* dst = d_value + (((long)b * Nv + lv.start[l] + tile_lo) * nH + head) * 64 + lane;
  const long pstride = (long)nH * 64;
#define MSDA_FLUSH(vec, base)                                                       \
  _Pragma("unroll") for (int e = 0; e < 32; ++e) {                                   \
    if ((base) + e < tile_n) {                                                       \
      if (nsplit == 1) dst[((base) + e) * pstride] = vec[e];                         \
      else atomicAdd(dst + ((base) + e) * pstride, vec[e]);                          \
    }                                                                                \
  }
  MSDA_FLUSH(a0, 0) MSDA_FLUSH(a1, 32) MSDA_FLUSH(a2, 64) MSDA_FLUSH(a3, 96)
#undef MSDA_FLUSH
#undef MSDA_DRAIN
}

static int msda_levels(const int* spatial_hw, int L, int Nv, MsdaLevels& lv) {
  if (L < 1 || L > MSDA_MAX_L) return GE_ERR_UNSUPPORTED;
  long start = 0;
  for (int l = 0; l < L; ++l) {
    lv.H[l] = spatial_hw[2 * l]; lv.W[l] = spatial_hw[2 * l + 1]; lv.start[l] = (int)start;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return GE_ERR_BAD_ARG;
    start += (long)lv.H[l] * lv.W[l];
  }
  return start == Nv ? GE_OK : GE_ERR_BAD_ARG;
}

extern "C" int ge_msda_fwd(const void* value, const int* spatial_hw, const float* loc, const float* attw, void* out,
                           int B, int Nv, int Nq, int nH, int L, int P, int dtype, void* stream) {
  if (!value || !spatial_hw || !loc || !attw || !out || B < 0 || Nv <= 0 || Nq < 0 || nH <= 0 || P <= 0) return GE_ERR_BAD_ARG;
  MsdaLevels lv;
  int e = msda_levels(spatial_hw, L, Nv, lv);
  if (e) return e;
  const long n_groups = (long)B * Nq * nH;
  if (n_groups == 0) return GE_OK;
  const unsigned blocks = ge_blocks(n_groups, 16, 1 << 22);
  if (dtype == GE_F32)
    msda_fwd_k<float><<<blocks, 256, 0, ge_stream(stream)>>>((const float*)value, lv, loc, attw, (float*)out, n_groups, Nv, Nq, nH, L, P);
  else if (dtype == GE_BF16)
    msda_fwd_k<bf16_t><<<blocks, 256, 0, ge_stream(stream)>>>((const bf16_t*)value, lv, loc, attw, (bf16_t*)out, n_groups, Nv, Nq, nH, L, P);
  else
    return GE_ERR_UNSUPPORTED;
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_msda_bwd(const void* value, const int* spatial_hw, const float* loc, const float* attw, const void* d_out,
                           float* d_value, float* d_loc, float* d_attw, int B, int Nv, int Nq, int nH, int L, int P,
                           int dtype, void* stream) {
  if (!value || !spatial_hw || !loc || !attw || !d_out || !d_value || !d_loc || !d_attw) return GE_ERR_BAD_ARG;
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
  // Default: one pass, d_value through coalesced 256-byte fp32 atomic bursts (at the L2 atomic-unit limit).
  // GE_MSDA_BWD=tile selects the experimental two-pass form: (1) d_loc / d_attw with reads only, (2) d_value by the
  // register tile-owner scatter below (correct, currently slower: kept for the next round's profiling).
  static const bool tile_mode = [] { const char* e = getenv("GE_MSDA_BWD"); return e && e[0] == 't'; }();
#define MSDA_BWD(TT, PP, AT)                                                                                               \
  msda_bwd_k<TT, PP, AT><<<blocks, 256, 0, s>>>((const TT*)value, lv, loc, attw, (const TT*)d_out, d_value, d_loc, d_attw, \
                                                n_groups, Nv, Nq, nH, L)
#define MSDA_BWD_P(TT, AT) do { if (P == 8) MSDA_BWD(TT, 8, AT); else MSDA_BWD(TT, 4, AT); } while (0)
  if (!tile_mode) {
    if (dtype == GE_F32) MSDA_BWD_P(float, true); else MSDA_BWD_P(bf16_t, true);
    GE_LAUNCH_CHECK();
    return GE_OK;
  }
  if (dtype == GE_F32) MSDA_BWD_P(float, false); else MSDA_BWD_P(bf16_t, false);
#undef MSDA_BWD_P
#undef MSDA_BWD
  GE_LAUNCH_CHECK();
  MsdaTileMap tm;
  int ntiles = 0, max_tiles = 1;
  for (int l = 0; l < L; ++l) max_tiles = std::max(max_tiles, (lv.H[l] * lv.W[l] + MSDA_TILE - 1) / MSDA_TILE);
  for (int l = 0; l < L; ++l) {
    const int tiles = (lv.H[l] * lv.W[l] + MSDA_TILE - 1) / MSDA_TILE;
    tm.splits[l] = std::max(1, std::min(std::min(128, Nq), (max_tiles + tiles / 2) / tiles));
    tm.first_block[l] = ntiles;
    ntiles += tiles * tm.splits[l];
  }
  tm.first_block[L] = ntiles;
  if (Nq >= (1 << 23)) return GE_ERR_UNSUPPORTED;                   // query index is packed into 24 bits
  dim3 grid((unsigned)((ntiles + 3) / 4), (unsigned)(B * nH));
  if (dtype == GE_F32)
    msda_bwd_value_k<float><<<grid, 256, 0, s>>>(lv, tm, loc, attw, (const float*)d_out, d_value, Nv, Nq, nH, L, P, ntiles);
  else
    msda_bwd_value_k<bf16_t><<<grid, 256, 0, s>>>(lv, tm, loc, attw, (const bf16_t*)d_out, d_value, Nv, Nq, nH, L, P, ntiles);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
