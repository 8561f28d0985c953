// Layout changes of the HAHI neck between feature maps (B, C, H*W) and token sequences (B, H*W, C), with the
// element-wise work that surrounds them in the reference folded into the same pass:
//   tokens_from_map : tok[b, n, c] = (map[b, c, n] + pos[c, n]) * drop(b, n, c)
//       = `conv_skip.flatten(2).transpose(1, 2)` + `query + query_pos`    (necks/hahi.py:303-306,316-318; mmcv
//         MultiScaleDeformableAttention.forward "query = query + query_pos"), and the backward of map_from_tokens;
//   map_from_tokens : map[b, c, n] = tok[b, n, c] * drop(b, n, c) + res[b, c, n]
//       = `self.dropout(output) + identity` followed by `fusion.permute(0, 2, 1).reshape(bs, c, h, w)` written
//         straight into its slot of the `torch.cat([fusion, feat_conv], dim=1)` buffer (necks/hahi.py:326-333,
//         338-346), and the backward of tokens_from_map.
// Both are HBM-bound: one read and one write of the tensor (plus pos / res), 64x64 tiles transposed through LDS
// with 16-byte global accesses on both sides.  Dropout uses a counter-based hash of (seed, logical element index)
// so the backward pass regenerates the mask instead of storing it.
#include "common.h"

#define NECK_TILE 64

// grid (ceil(N/64), ceil(C/64), B), 256 threads.  VEC: N and C multiples of the 16-byte vector, all strides aligned.
template <typename T, bool VEC, bool DROP>
__global__ void __launch_bounds__(256) tokens_from_map_k(const T* __restrict__ map, long map_bs, const float* __restrict__ pos,
                                                         T* __restrict__ tok, long tok_bs, int C, long N, float p, float inv_keep,
                                                         uint64_t seed0, const unsigned long long* __restrict__ salt) {
  const uint64_t seed = ge_salted(seed0, salt);
  constexpr int VN = V8<T>::N, LPR = NECK_TILE / VN, RPP = 256 / LPR;
  __shared__ float tile[NECK_TILE][NECK_TILE + 1];              // [c][n]
  const long n0 = (long)blockIdx.x * NECK_TILE;
  const int c0 = blockIdx.y * NECK_TILE, b = blockIdx.z;
  const int lane_v = (threadIdx.x % LPR) * VN, row0 = threadIdx.x / LPR;
  const T* mp = map + (long)b * map_bs;
#pragma unroll
  for (int r = row0; r < NECK_TILE; r += RPP) {
    const int c = c0 + r;
    const long n = n0 + lane_v;
    float v[VN];
    if (VEC) {
      if (c < C && n < N) {
        V8<T>::ld(mp + (long)c * N + n, v);
        if (pos) {
#pragma unroll
          for (int k = 0; k < VN; k += 4) {
            const float4 q = *(const float4*)(pos + (long)c * N + n + k);
            v[k] += q.x; v[k + 1] += q.y; v[k + 2] += q.z; v[k + 3] += q.w;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        v[k] = 0.f;
        if (c < C && n + k < N) v[k] = Io<T>::ld(mp + (long)c * N + n + k) + (pos ? pos[(long)c * N + n + k] : 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) tile[r][lane_v + k] = v[k];
  }
  __syncthreads();
  T* tp = tok + (long)b * tok_bs;
#pragma unroll
  for (int r = row0; r < NECK_TILE; r += RPP) {
    const long n = n0 + r;
    const int c = c0 + lane_v;
    if (n >= N || c >= C) continue;
    float v[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = tile[lane_v + k][r];
    if (DROP) {
      const uint64_t base = ((uint64_t)b * (uint64_t)N + (uint64_t)n) * (uint64_t)C + (uint64_t)c;
#pragma unroll
      for (int k = 0; k < VN; ++k) v[k] *= ge_drop_scale(seed, base + k, ge_drop_threshold(p), inv_keep);
    }
    if (VEC) {
      V8<T>::st(tp + n * C + c, v);
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k)
        if (c + k < C) Io<T>::st(tp + n * C + c + k, v[k]);
    }
  }
}

template <typename T, bool VEC, bool DROP>
__global__ void __launch_bounds__(256) map_from_tokens_k(const T* __restrict__ tok, long tok_bs, const T* __restrict__ res, long res_bs,
                                                         T* __restrict__ map, long map_bs, int C, long N, float p, float inv_keep,
                                                         uint64_t seed0, const unsigned long long* __restrict__ salt) {
  const uint64_t seed = ge_salted(seed0, salt);
  constexpr int VN = V8<T>::N, LPR = NECK_TILE / VN, RPP = 256 / LPR;
  __shared__ float tile[NECK_TILE][NECK_TILE + 1];              // [c][n]
  const long n0 = (long)blockIdx.x * NECK_TILE;
  const int c0 = blockIdx.y * NECK_TILE, b = blockIdx.z;
  const int lane_v = (threadIdx.x % LPR) * VN, row0 = threadIdx.x / LPR;
  const T* tp = tok + (long)b * tok_bs;
#pragma unroll
  for (int r = row0; r < NECK_TILE; r += RPP) {
    const long n = n0 + r;
    const int c = c0 + lane_v;
    float v[VN];
    if (VEC) {
      if (n < N && c < C) {
        V8<T>::ld(tp + n * C + c, v);
      } else {
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k) v[k] = (n < N && c + k < C) ? Io<T>::ld(tp + n * C + c + k) : 0.f;
    }
    if (DROP) {
      const uint64_t base = ((uint64_t)b * (uint64_t)N + (uint64_t)n) * (uint64_t)C + (uint64_t)c;
#pragma unroll
      for (int k = 0; k < VN; ++k) v[k] *= ge_drop_scale(seed, base + k, ge_drop_threshold(p), inv_keep);
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) tile[lane_v + k][r] = v[k];
  }
  __syncthreads();
  T* mp = map + (long)b * map_bs;
  const T* rp = res ? res + (long)b * res_bs : nullptr;
#pragma unroll
  for (int r = row0; r < NECK_TILE; r += RPP) {
    const int c = c0 + r;
    const long n = n0 + lane_v;
    if (c >= C || n >= N) continue;
    float v[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) v[k] = tile[r][lane_v + k];
    if (VEC) {
      if (rp) {
        float q[VN];
        V8<T>::ld(rp + (long)c * N + n, q);
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] += q[k];
      }
      V8<T>::st(mp + (long)c * N + n, v);
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k)
        if (n + k < N) Io<T>::st(mp + (long)c * N + n + k, v[k] + (rp ? Io<T>::ld(rp + (long)c * N + n + k) : 0.f));
    }
  }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static const unsigned long long* g_rng_salt = nullptr;
const unsigned long long* ge_rng_salt_get() { return g_rng_salt; }
extern "C" int ge_rng_salt(const unsigned long long* device_counter) { g_rng_salt = device_counter; return GE_OK; }

template <typename T>
static int tokens_from_map_launch(const void* map, long map_bs, const float* pos, void* tok, long tok_bs, int B, int C, long N,
                                  float p, uint64_t seed, hipStream_t s) {
  constexpr int VN = V8<T>::N;
  const bool vec = N % VN == 0 && C % VN == 0 && map_bs % VN == 0 && tok_bs % VN == 0 && al16(map) && al16(tok) && (!pos || al16(pos));
  const bool drop = p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p) : 1.f;
  dim3 grid((unsigned)((N + NECK_TILE - 1) / NECK_TILE), (unsigned)((C + NECK_TILE - 1) / NECK_TILE), (unsigned)B);
#define GE_TFM(V, D) tokens_from_map_k<T, V, D><<<grid, 256, 0, s>>>((const T*)map, map_bs, pos, (T*)tok, tok_bs, C, N, p, inv_keep, seed, ge_rng_salt_get())
  if (vec) { if (drop) GE_TFM(true, true); else GE_TFM(true, false); }
  else { if (drop) GE_TFM(false, true); else GE_TFM(false, false); }
#undef GE_TFM
  GE_LAUNCH_CHECK();
  return GE_OK;
}
template <typename T>
static int map_from_tokens_launch(const void* tok, long tok_bs, const void* res, long res_bs, void* map, long map_bs, int B, int C,
                                  long N, float p, uint64_t seed, hipStream_t s) {
  constexpr int VN = V8<T>::N;
  const bool vec = N % VN == 0 && C % VN == 0 && map_bs % VN == 0 && tok_bs % VN == 0 && al16(map) && al16(tok) &&
                   (!res || (al16(res) && res_bs % VN == 0));
  const bool drop = p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p) : 1.f;
  dim3 grid((unsigned)((N + NECK_TILE - 1) / NECK_TILE), (unsigned)((C + NECK_TILE - 1) / NECK_TILE), (unsigned)B);
#define GE_MFT(V, D) map_from_tokens_k<T, V, D><<<grid, 256, 0, s>>>((const T*)tok, tok_bs, (const T*)res, res_bs, (T*)map, map_bs, C, N, p, inv_keep, seed, ge_rng_salt_get())
  if (vec) { if (drop) GE_MFT(true, true); else GE_MFT(true, false); }
  else { if (drop) GE_MFT(false, true); else GE_MFT(false, false); }
#undef GE_MFT
  GE_LAUNCH_CHECK();
  return GE_OK;
}

static inline bool neck_args_ok(int B, int C, long N, float p) {
  return B >= 0 && B <= 65535 && C > 0 && N >= 0 && (N + NECK_TILE - 1) / NECK_TILE <= 2147483647L && p >= 0.f && p < 1.f;
}

extern "C" int ge_tokens_from_map(const void* map, long map_bs, const float* pos, void* tok, long tok_bs, int B, int C, long N,
                                  float p_drop, unsigned long long seed, int dtype, void* stream) {
  if (!map || !tok || !neck_args_ok(B, C, N, p_drop)) return GE_ERR_BAD_ARG;
  if ((long)B * C * N == 0) return GE_OK;
  if (dtype == GE_F32) return tokens_from_map_launch<float>(map, map_bs, pos, tok, tok_bs, B, C, N, p_drop, seed, ge_stream(stream));
  if (dtype == GE_BF16) return tokens_from_map_launch<bf16_t>(map, map_bs, pos, tok, tok_bs, B, C, N, p_drop, seed, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

extern "C" int ge_map_from_tokens(const void* tok, long tok_bs, const void* res, long res_bs, void* map, long map_bs, int B, int C,
                                  long N, float p_drop, unsigned long long seed, int dtype, void* stream) {
  if (!map || !tok || !neck_args_ok(B, C, N, p_drop)) return GE_ERR_BAD_ARG;
  if ((long)B * C * N == 0) return GE_OK;
  if (dtype == GE_F32) return map_from_tokens_launch<float>(tok, tok_bs, res, res_bs, map, map_bs, B, C, N, p_drop, seed, ge_stream(stream));
  if (dtype == GE_BF16) return map_from_tokens_launch<bf16_t>(tok, tok_bs, res, res_bs, map, map_bs, B, C, N, p_drop, seed, ge_stream(stream));
  return GE_ERR_UNSUPPORTED;
}

// ============================================================================ residual + stochastic depth
// out[b, i] = identity[b, i] + branch[b, i] * scale[b]      (scale[b] = Bernoulli(keep) / keep per sample)
// = `identity + DropPath(branch)` of the Swin blocks (depthformer_swin.py:461-472 via mmcv DropPath / FFN): ATen runs a
// divide, a multiply and an add over the token tensor; this is one pass, mixed precision (fp32 residual stream + bf16
// branch in stage 0).  Backward of the branch: d_branch = d_out * scale[b] (ge_scale_rows); the identity gradient is d_out.
template <typename T> struct E8;       // 8 consecutive elements <-> 8 floats
template <> struct E8<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[8]) { V8<float>::ld(p, v); V8<float>::ld(p + 4, v + 4); }
  static __device__ __forceinline__ void st(float* p, const float v[8]) { V8<float>::st(p, v); V8<float>::st(p + 4, v + 4); }
};
template <> struct E8<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[8]) { V8<bf16_t>::ld(p, v); }
  static __device__ __forceinline__ void st(bf16_t* p, const float v[8]) { V8<bf16_t>::st(p, v); }
};

// HAS_ID = false: out = branch * scale (the backward pass)
template <typename TI, typename TB, typename TO, bool HAS_ID, bool VEC>
__global__ void __launch_bounds__(256) scale_add_k(const TI* __restrict__ identity, const TB* __restrict__ branch,
                                                   const float* __restrict__ scale, TO* __restrict__ out, long per_sample) {
  const int b = blockIdx.y;
  const float sc = scale[b];
  const long base = (long)b * per_sample;
  if (VEC) {
    const long nv = per_sample / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float x[8], y[8];
      E8<TB>::ld(branch + base + i * 8, y);
      if (HAS_ID) {
        E8<TI>::ld(identity + base + i * 8, x);
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = x[k] + y[k] * sc;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = y[k] * sc;
      }
      E8<TO>::st(out + base + i * 8, y);
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long)gridDim.x * blockDim.x) {
      const float y = Io<TB>::ld(branch + base + i) * sc;
      Io<TO>::st(out + base + i, HAS_ID ? Io<TI>::ld(identity + base + i) + y : y);
    }
  }
}

template <typename TI, typename TB, typename TO, bool HAS_ID>
static int scale_add_launch(const void* identity, const void* branch, const float* scale, void* out, int B, long per_sample,
                            hipStream_t s) {
  const bool vec = per_sample % 8 == 0 && al16(branch) && al16(out) && (!HAS_ID || al16(identity));
  long gx = ((vec ? per_sample / 8 : per_sample) + 256 * 4 - 1) / (256 * 4);
  if (gx < 1) gx = 1;
  if (gx > 4096) gx = 4096;
  dim3 grid((unsigned)gx, (unsigned)B);
  if (vec) scale_add_k<TI, TB, TO, HAS_ID, true><<<grid, 256, 0, s>>>((const TI*)identity, (const TB*)branch, scale, (TO*)out, per_sample);
  else scale_add_k<TI, TB, TO, HAS_ID, false><<<grid, 256, 0, s>>>((const TI*)identity, (const TB*)branch, scale, (TO*)out, per_sample);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

extern "C" int ge_residual_scale_add(const void* identity, int id_dtype, const void* branch, int br_dtype, const float* scale,
                                     void* out, int B, long per_sample, void* stream) {
  if (!identity || !branch || !scale || !out || B < 0 || B > 65535 || per_sample < 0) return GE_ERR_BAD_ARG;
  if ((long)B * per_sample == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  if (id_dtype == GE_F32 && br_dtype == GE_F32) return scale_add_launch<float, float, float, true>(identity, branch, scale, out, B, per_sample, s);
  if (id_dtype == GE_F32 && br_dtype == GE_BF16) return scale_add_launch<float, bf16_t, float, true>(identity, branch, scale, out, B, per_sample, s);
  if (id_dtype == GE_BF16 && br_dtype == GE_BF16) return scale_add_launch<bf16_t, bf16_t, bf16_t, true>(identity, branch, scale, out, B, per_sample, s);
  return GE_ERR_UNSUPPORTED;
}

extern "C" int ge_scale_rows(const void* x, int x_dtype, const float* scale, void* out, int out_dtype, int B, long per_sample,
                             void* stream) {
  if (!x || !scale || !out || B < 0 || B > 65535 || per_sample < 0) return GE_ERR_BAD_ARG;
  if ((long)B * per_sample == 0) return GE_OK;
  hipStream_t s = ge_stream(stream);
  if (x_dtype == GE_F32 && out_dtype == GE_F32) return scale_add_launch<float, float, float, false>(nullptr, x, scale, out, B, per_sample, s);
  if (x_dtype == GE_F32 && out_dtype == GE_BF16) return scale_add_launch<float, float, bf16_t, false>(nullptr, x, scale, out, B, per_sample, s);
  if (x_dtype == GE_BF16 && out_dtype == GE_BF16) return scale_add_launch<bf16_t, bf16_t, bf16_t, false>(nullptr, x, scale, out, B, per_sample, s);
  if (x_dtype == GE_BF16 && out_dtype == GE_F32) return scale_add_launch<bf16_t, bf16_t, float, false>(nullptr, x, scale, out, B, per_sample, s);
  return GE_ERR_UNSUPPORTED;
}
