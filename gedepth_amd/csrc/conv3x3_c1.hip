// 3x3 / stride 1 / pad 1 convolution with ONE output channel on channels-last bf16 maps: the depth regressor of the decode head
// (reference depth/models/decode_heads/decode_head.py: `conv_depth = nn.Conv2d(channels, 1, 3, padding=1)`) and the ground-attention
// head of the PE neck (necks/pemask_neck.py:36-42 `convfinal`).  As an implicit GEMM this layer has N = 1: the libraries run it at
// 7 TFLOP/s and 0.75 TB/s (profiles/r3_library_roofline.txt: 0.28 ms forward, 0.49 ms backward per step); it is a streaming
// reduction — 9 FLOP per byte — and belongs on the vector pipe at HBM speed.
//
// Lane layout: 8 lanes own one pixel (16 bytes = 8 channels each, 64 channels per step), a wave 8 consecutive pixels of the linear
// (n, y, x) order: every tap is one 1 KB contiguous read per wave, the nine taps of a pixel hit L1 / L2 (three image rows are live).
// The weights sit in LDS as bf16 pairs — rounded exactly as autocast's weight cast does — and meet the activations in
// v_dot2c_f32_bf16 (two channels per instruction, fp32 accumulation).
//
// Backward is ONE pass for all three gradients: with d_t = dy[p - off(t)] (nine scalars per pixel)
//     dx[p, :]  = sum_t d_t w[t, :]              (data gradient, written once)
//     dw[t, :] += d_t x[p, :]                    (weight gradient: x is only needed at the CENTRE pixel)
//     db       += d_4                            (bias gradient)
// so x is read once and dx written once; the 9 x 8 partial dw of a lane stay in registers over the whole pixel loop and are flushed
// with one fp32 atomic per (tap, channel) and workgroup.
#include "common.h"

typedef __bf16 c1_bf2 __attribute__((ext_vector_type(2)));

#define C1_THREADS 256
#define C1_MAX_CIN 1024

__device__ __forceinline__ float c1_dot8(const uint4 a, const uint4 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(c1_bf2, a.x), __builtin_bit_cast(c1_bf2, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(c1_bf2, a.y), __builtin_bit_cast(c1_bf2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(c1_bf2, a.z), __builtin_bit_cast(c1_bf2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(c1_bf2, a.w), __builtin_bit_cast(c1_bf2, b.w), acc, false);
  return acc;
}

template <typename TO> struct C1Out;
template <> struct C1Out<float> { static __device__ __forceinline__ void st(float* p, float v) { *p = v; } static __device__ __forceinline__ float ld(const float* p) { return *p; } };
template <> struct C1Out<bf16_t> {
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
};

// Work decomposition (both directions): a unit = 4 image rows x C1_XSEG pixels; wave w of the workgroup owns row w of the unit and walks it 8
// pixels at a time, so the vertical taps of a wave are the rows its neighbours in the same CU load (L1), and no per-pixel divisions are
// needed.  Units are dealt to workgroups XCD by XCD in contiguous ranges (blockIdx % 8 = XCD): the halo rows two units share stay in ONE L2.
#define C1_XSEG 128
#define C1_XCDS 8

struct C1Units { int nxs, rows_q, total, per_xcd; };
__device__ __forceinline__ bool c1_unit(const C1Units& un, int u, int H, int wv, int& n, int& y, int& xs) {
  xs = u % un.nxs;
  const int rq = u / un.nxs;
  n = rq / un.rows_q;
  y = (rq - n * un.rows_q) * 4 + wv;
  return y < H;
}

template <typename TO>
__global__ void __launch_bounds__(C1_THREADS) conv3x3_c1_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               TO* __restrict__ y, C1Units un, int H, int W, int Cin) {
  __shared__ __attribute__((aligned(16))) bf16_t wl[9 * C1_MAX_CIN];               // [tap][Cin] bf16
  for (int i = threadIdx.x; i < 9 * Cin; i += C1_THREADS) wl[i] = f2bf(w[i]);
  __syncthreads();
  const int sub = threadIdx.x & 7, slot = (threadIdx.x & 63) >> 3, wv = threadIdx.x >> 6;
  const float b0 = bias ? bias[0] : 0.f;
  const int xcd = blockIdx.x % C1_XCDS, u_end = min(un.total, (xcd + 1) * un.per_xcd);
  for (int u = xcd * un.per_xcd + blockIdx.x / C1_XCDS; u < u_end; u += gridDim.x / C1_XCDS) {
    int n, yh, xs;
    if (!c1_unit(un, u, H, wv, n, yh, xs)) continue;
    const int x_end = min(W, (xs + 1) * C1_XSEG);
    const long row = ((long)n * H + yh) * W;
    for (int xw = xs * C1_XSEG + slot; xw < x_end; xw += 8) {
      const long p = row + xw;
      float acc = 0.f;
      for (int c0 = sub * 8; c0 < Cin; c0 += 64) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int r = t / 3, s = t - 3 * r;
          const int yy = yh + r - 1, xx = xw + s - 1;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *(const uint4*)(x + (p + (long)(r - 1) * W + (s - 1)) * Cin + c0);
          acc = c1_dot8(v, *(const uint4*)(wl + t * Cin + c0), acc);
        }
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (sub == 0) C1Out<TO>::st(y + p, acc + b0);
    }
  }
}

// One chunk of 64 input channels per outer iteration (Cin = 64 on the GEDepth path: a single pass over the pixels).  The dy halo tile
// of a unit (6 rows x (C1_XSEG + 2) scalars) is staged in LDS as fp32: the nine d_t of a pixel are LDS reads, not nine 2-byte gathers.
template <typename TD>
__global__ void __launch_bounds__(C1_THREADS) conv3x3_c1_bwd_k(const bf16_t* __restrict__ x, const TD* __restrict__ dy, const float* __restrict__ w,
                                                               bf16_t* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, C1Units un,
                                                               int H, int W, int Cin) {
  __shared__ __attribute__((aligned(16))) float red[4 * 9 * 64];
  __shared__ __attribute__((aligned(16))) float wl[9 * 64];                        // the chunk's weights, bf16-rounded, [tap][64]
  __shared__ float dyt[6 * (C1_XSEG + 2)];
  const int sub = threadIdx.x & 7, lane = threadIdx.x & 63, slot = lane >> 3, wv = threadIdx.x >> 6;
  const int xcd = blockIdx.x % C1_XCDS, u_end = min(un.total, (xcd + 1) * un.per_xcd);
  float dbs = 0.f;
  for (int cb = 0; cb < Cin; cb += 64) {
    const int c0 = cb + sub * 8;
    const bool live = c0 < Cin;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += C1_THREADS) {
      const int t = i >> 6, c = cb + (i & 63);
      wl[i] = c < Cin ? bf2f(f2bf(w[t * Cin + c])) : 0.f;
    }
    for (int u = xcd * un.per_xcd + blockIdx.x / C1_XCDS; u < u_end; u += gridDim.x / C1_XCDS) {
      const int xs = u % un.nxs, rq = u / un.nxs, n = rq / un.rows_q, y0 = (rq - n * un.rows_q) * 4, x0 = xs * C1_XSEG;
      __syncthreads();                                                              // previous unit's readers are done (and wl is visible)
      for (int i = threadIdx.x; i < 6 * (C1_XSEG + 2); i += C1_THREADS) {
        const int ry = i / (C1_XSEG + 2), rx = i - ry * (C1_XSEG + 2);
        const int yy = y0 + ry - 1, xx = x0 + rx - 1;
        float d = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) d = C1Out<TD>::ld(dy + ((long)n * H + yy) * W + xx);
        dyt[i] = d;
      }
      __syncthreads();
      const int yh = y0 + wv;
      if (yh >= H) continue;
      const int x_end = min(W, x0 + C1_XSEG);
      const long row = ((long)n * H + yh) * W;
      for (int xw = x0 + slot; xw < x_end; xw += 8) {
        const long p = row + xw;
        float xv[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[j] = 0.f; o[j] = 0.f; }
        if (live) V8<bf16_t>::ld(x + p * Cin + c0, xv);
        const float* dc = dyt + (wv + 1) * (C1_XSEG + 2) + (xw - x0) + 1;           // dy at this pixel
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int r = t / 3, s = t - 3 * r;
          const float d = dc[(1 - r) * (C1_XSEG + 2) + (1 - s)];                    // the output pixel whose tap t reads input pixel p
          if (t == 4 && cb == 0 && sub == 0) dbs += d;
          const float4 w0 = *(const float4*)(wl + t * 64 + sub * 8), w1 = *(const float4*)(wl + t * 64 + sub * 8 + 4);
          const float wr[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) { o[j] = fmaf(d, wr[j], o[j]); acc[t][j] = fmaf(d, xv[j], acc[t][j]); }
        }
        if (live) V8<bf16_t>::st(dx + p * Cin + c0, o);
      }
    }
    // sum over the 8 pixel slots of the wave, then over the 4 waves, then one atomic per (tap, channel)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[t][j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8) red[(wv * 9 + t) * 64 + sub * 8 + j] = v;
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += C1_THREADS) {
      const int t = i >> 6, c = cb + (i & 63);
      if (c < Cin) atomicAdd(dw + t * Cin + c, red[i] + red[9 * 64 + i] + red[2 * 9 * 64 + i] + red[3 * 9 * 64 + i]);
    }
  }
  if (db) {
    dbs += __shfl_xor(dbs, 8, 64);
    dbs += __shfl_xor(dbs, 16, 64);
    dbs += __shfl_xor(dbs, 32, 64);
    if (lane == 0) atomicAdd(db, dbs);
  }
}

static inline C1Units c1_units(int N, int H, int W) {
  C1Units un;
  un.nxs = (W + C1_XSEG - 1) / C1_XSEG;
  un.rows_q = (H + 3) / 4;
  un.total = N * un.rows_q * un.nxs;
  un.per_xcd = (un.total + C1_XCDS - 1) / C1_XCDS;
  return un;
}
static inline unsigned c1_grid(const C1Units& un, int cap) {          // a multiple of the XCD count: blockIdx % 8 picks the XCD's unit range
  int per = un.per_xcd < cap / C1_XCDS ? un.per_xcd : cap / C1_XCDS;
  return (unsigned)((per < 1 ? 1 : per) * C1_XCDS);
}

// y (N, H, W) [out_dtype: GE_BF16 or GE_F32] = bias + conv3x3(x (N, H, W, Cin) bf16 channels-last, w (1, 3, 3, Cin) fp32, rounded to bf16 as
// autocast's weight cast does).  Cin % 8 == 0, Cin <= 1024.
extern "C" int ge_conv3x3_c1_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin, int out_dtype, void* stream) {
  if (!x || !w || !y || N < 0 || H <= 0 || W <= 0 || Cin <= 0) return GE_ERR_BAD_ARG;
  if (Cin % 8 || Cin > C1_MAX_CIN || (out_dtype != GE_BF16 && out_dtype != GE_F32) || ((uintptr_t)x & 15)) return GE_ERR_UNSUPPORTED;
  if ((long)N * ((H + 3) / 4) * ((W + C1_XSEG - 1) / C1_XSEG) > 0x7fffffffL) return GE_ERR_UNSUPPORTED;
  if (N == 0) return GE_OK;
  const C1Units un = c1_units(N, H, W);
  const unsigned g = c1_grid(un, 256 * 16);
  if (out_dtype == GE_F32) conv3x3_c1_fwd_k<float><<<g, C1_THREADS, 0, ge_stream(stream)>>>((const bf16_t*)x, w, bias, (float*)y, un, H, W, Cin);
  else conv3x3_c1_fwd_k<bf16_t><<<g, C1_THREADS, 0, ge_stream(stream)>>>((const bf16_t*)x, w, bias, (bf16_t*)y, un, H, W, Cin);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// dx (N, H, W, Cin) bf16, dw (1, 3, 3, Cin) fp32 and db (1) fp32 [may be NULL] from dy (N, H, W) [dy_dtype]; dw / db are zero-filled here.
extern "C" int ge_conv3x3_c1_bwd(const void* x, const void* dy, const float* w, void* dx, float* dw, float* db, int N, int H, int W, int Cin,
                                 int dy_dtype, void* stream) {
  if (!x || !dy || !w || !dx || !dw || N < 0 || H <= 0 || W <= 0 || Cin <= 0) return GE_ERR_BAD_ARG;
  if (Cin % 8 || Cin > C1_MAX_CIN || (dy_dtype != GE_BF16 && dy_dtype != GE_F32) || (((uintptr_t)x | (uintptr_t)dx) & 15)) return GE_ERR_UNSUPPORTED;
  hipStream_t s = ge_stream(stream);
  hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * 9 * Cin, s);
  if (e != hipSuccess) return (int)e;
  if (db && (e = hipMemsetAsync(db, 0, sizeof(float), s)) != hipSuccess) return (int)e;
  if ((long)N * ((H + 3) / 4) * ((W + C1_XSEG - 1) / C1_XSEG) > 0x7fffffffL) return GE_ERR_UNSUPPORTED;
  if (N == 0) return GE_OK;
  const C1Units un = c1_units(N, H, W);
  const unsigned g = c1_grid(un, 256 * 4);               // persistent: a lane's 72 partial dw cover several units before the atomic flush
  if (dy_dtype == GE_F32) conv3x3_c1_bwd_k<float><<<g, C1_THREADS, 0, s>>>((const bf16_t*)x, (const float*)dy, w, (bf16_t*)dx, dw, db, un, H, W, Cin);
  else conv3x3_c1_bwd_k<bf16_t><<<g, C1_THREADS, 0, s>>>((const bf16_t*)x, (const bf16_t*)dy, w, (bf16_t*)dx, dw, db, un, H, W, Cin);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
