// Shared device helpers for the gfx950 kernels of libgedepth_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gedepth_hip.h"

#define GE_WAVE 64

#define GE_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, like torch
#ifndef GE_SW_BF16
  // gfx950's conversion unit (v_cvt_pk_bf16_f32): same rounding, NaN stays NaN.  The integer sequence below costs ~6 VALU per value — 48 per 16-byte
  // store of the streaming kernels: step 47.47 -> 47.12 ms same-session, conv1x1_bn_act_k 505 -> 391 us (round 5); -DGE_SW_BF16 restores it (A/B)
  return __builtin_bit_cast(bf16_t, (__bf16)f);
#endif
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rt(float v) { return v; }   // round-trip through storage type
};
template <> struct Io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float rt(float v) { return bf2f(f2bf(v)); }
};

// 16-byte vectors of the storage type, widened to fp32 registers (8 bf16 or 4 f32 per lane and instruction)
template <typename T> struct V8;
template <> struct V8<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[8]) {
    const uint4 t = *(const uint4*)p; const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float v[8]) {
    uint4 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    t.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); t.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *(uint4*)p = t;
  }
};
template <> struct V8<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float v[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};

// F.interpolate(mode='bilinear') source-index rule (scale from sizes, not from scale_factor).
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ float ge_scale(int in, int out, bool align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __forceinline__ Lerp ge_lerp(int dst, int in, float scale, bool align) {
  float src = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  Lerp r;
  r.i0 = (int)src;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.w1 = src - (float)r.i0;
  r.w0 = 1.f - r.w1;
  return r;
}

static inline hipStream_t ge_stream(void* s) { return (hipStream_t)s; }

// Compute-unit count of the CURRENT device, cached per device id (a process may drive several devices; a single cached value would size the
// persistent grids of every device after the first caller's).  0 on error.
static inline int ge_cu_count() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && __atomic_load_n(&cache[dev], __ATOMIC_RELAXED)) return cache[dev];
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  if (dev < 64) __atomic_store_n(&cache[dev], p.multiProcessorCount, __ATOMIC_RELAXED);
  return p.multiProcessorCount;
}

// Dropout masks are a counter-based hash of (seed, element index) with the seed a launch ARGUMENT — frozen when the launch is captured in a
// hipGraph.  ge_rng_salt(ptr) (neck.hip) registers a device counter that every dropout kernel adds to its seed at EXECUTION time; a
// captured step increments it inside the graph, so each replay draws fresh masks (forward and backward of one step read the same value).
const unsigned long long* ge_rng_salt_get();
__device__ __forceinline__ uint64_t ge_salted(uint64_t seed, const unsigned long long* salt) {
  return salt ? seed + (uint64_t)*salt * 0x9E3779B97F4A7C15ull : seed;
}
// Dropout keep decisions: ONE 64-bit hash (splitmix64 finaliser) per group of four consecutive element indices, 16 bits per element:
// element idx is kept when bits [16 (idx & 3), +16) of hash(seed, idx >> 2) are >= thr = round(p * 65536) (keep probability 1 - thr / 65536: p = 0.1 ->
// 0.899994).  The per-element 32-bit hash of rounds 2 - 4 (seven multiplies and a dozen shifts / xors per value) cost the cross-attention's concat /
// slice passes 0.22 ms per step; the streaming kernels take whole groups (ge_drop_scale4), the transposing ones single elements.
__device__ __forceinline__ uint64_t ge_drop_hash4(uint64_t seed, uint64_t group) {
  uint64_t z = group * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t ge_drop_threshold(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
__device__ __forceinline__ float ge_drop_scale(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
  const uint32_t u = (uint32_t)(ge_drop_hash4(seed, idx >> 2) >> (16 * (unsigned)(idx & 3))) & 0xffffu;
  return u >= thr ? inv_keep : 0.f;
}
__device__ __forceinline__ void ge_drop_scale4(uint64_t seed, uint64_t group, uint32_t thr, float inv_keep, float s[4]) {
  const uint64_t h = ge_drop_hash4(seed, group);
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = (((uint32_t)(h >> (16 * e))) & 0xffffu) >= thr ? inv_keep : 0.f;
}
static inline unsigned ge_blocks(long n, int per_block, long cap = 1 << 20) {
  long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}
