// 1x1 convolution + training-mode BatchNorm2d + (Leaky)ReLU [+ positional embedding] of a 64-channel channels-last map in ONE pass over
// the output (gfx950, bf16 storage) — the conv_proj block of the HAHI neck and the query it feeds to the cross-attention
// (depth/models/necks/hahi.py:151-157 ConvModule(in_channels[0], embedding_dim, 1, norm BN, act ReLU), :294-306 query = flatten(conv_skip) + pos).
//
// The output is 8 x wider than the input (64 -> 512 channels at 176 x 560: 0.1 GB in, 0.8 GB out per tensor), and the composition
// conv -> statistics pass -> normalise + ReLU pass -> position add moved that 0.8 GB seven times.  Here:
//   * the BatchNorm batch statistics of the convolution output z = x W^T follow from the first two moments of the INPUT:
//         mean_c = w_c . m        E[z_c^2] = w_c^T S w_c        m = sum x / n,  S = sum x x^T / n   (the 64 x 64 Gram matrix)
//     conv1x1_gram_k reads the 0.1 GB input once (MFMA on transposing LDS reads), conv1x1_bn_finalize_k turns S into the per-channel scale /
//     shift (and the running statistics, save_mean, save_rstd) — the pre-BN tensor z is never written;
//   * conv1x1_bn_act_k: y = act(a_c (x . w_c) + b_c) with the weights resident in registers (a wave owns 128 output channels), epilogue through a
//     wave-private LDS tile -> full 256-byte row pieces; the same pass writes query = y + pos (one more store instead of a read + write pass);
//   * backward (no pre-BN tensor exists): g = (dy_identity + d_query) * act'(y) in one pass with its column sums (conv1x1_bn_mask_k); with
//     G = g^T X (ge_conv1x1_nhwc_wgrad, fp32) everything else is small-matrix algebra on G, the column sums and S (conv1x1_bn_bwd_finalize_k):
//         d_beta = sum g,   d_gamma_c = rstd_c w_c . (G_c - sum g_c m)
//         dW_c   = a_c (G_c - s m1_c - u_c m2_c)           u_c = rstd_c (S' w_c - s mean_c), S' = sum x x^T, s = sum x, m1 = d_beta / n, m2 = d_gamma / n
//         dX     = g A1 + X A2 + c0       A1 = diag(a) W,  A2 = - W^T diag(rstd a m2) W,  c0 = ((mean rstd m2 - m1) a)^T W   (conv1x1_bn_dgrad_k:
//                                                 one pass over g and X with [A1; A2] resident in LDS)
//     (the BatchNorm backward's mean-subtraction terms are rank-64 corrections of the two GEMMs the convolution backward runs anyway).
// Statistics are those of the exact fp32 products (the two-pass path takes them from the bf16-rounded convolution output): bf16 mode only; the
// fp32 parity mode keeps the two-pass kernels (csrc/nhwc.hip).
#include "common.h"

typedef __bf16 cb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cb_bf16x4 __attribute__((ext_vector_type(4)));
typedef float cb_f32x16 __attribute__((ext_vector_type(16)));
#define CB_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

#define CB_K 64                         // input channels
#define CB_LD 72                        // LDS row stride (bf16) of a staged [token][64] tile: 144 B, 8-byte aligned pieces for the transposing reads
#define CB_GRAM (65 * CB_K)             // 64 x 64 sums of products + 64 column sums
#define CB_GRAM_WG 256                  // partial Gram matrices (one per workgroup)

// fp32 -> bf16 on the conversion unit (v_cvt_pk_bf16_f32, round to nearest even): the software rounding of common.h costs ~6 VALU per value,
// a third of this file's forward kernel
typedef __bf16 cb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t cb_bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t cb_pk(float lo, float hi) { return __builtin_bit_cast(uint32_t, (cb_bf16x2){(__bf16)lo, (__bf16)hi}); }

__device__ __forceinline__ int cb_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }      // C/D layout of v_mfma_f32_32x32x16: row of register r

// ------------------------------------------------------------------------------------------------ input moments
__global__ void __launch_bounds__(256) conv1x1_gram_k(const bf16_t* __restrict__ x, long rows, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) bf16_t stage_all[4][64 * CB_LD];
  __shared__ float red[CB_GRAM];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hi = lane >> 5;
  bf16_t* stage = stage_all[wv];
  cb_f32x16 acc[2][2], sum[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; sum[0][i] = 0.f; sum[1][i] = 0.f; }
  const cb_bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
  // transposing read: lane (n = lane & 31, kg = lane >> 5) ends up with rows 8 kg .. 8 kg + 7 of column n of a [16 rows][32 columns] block
  const int tr_row = hi * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const long nblk = (rows + 63) / 64;
  for (long blk = (long)blockIdx.x * 4 + wv; blk < nblk; blk += (long)gridDim.x * 4) {
    uint4 R[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane;
      const long row = blk * 64 + (idx >> 3);
      R[i] = row < rows ? *(const uint4*)(x + row * CB_K + (idx & 7) * 8) : make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane;
      *(uint4*)(stage + (idx >> 3) * CB_LD + (idx & 7) * 8) = R[i];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      cb_bf16x8 F[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const bf16_t* p = stage + (ks * 16 + tr_row) * CB_LD + cb * 32 + tr_col;
        const cb_bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(CB_LDS(cb_bf16x4, p));
        const cb_bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(CB_LDS(cb_bf16x4, p + 4 * CB_LD));
        F[cb] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[a], F[b], acc[a][b], 0, 0, 0);
        sum[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[a], ones, sum[a], 0, 0, 0);
      }
    }
  }
  for (int i = threadIdx.x; i < CB_GRAM; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(a * 32 + cb_row(r, hi)) * CB_K + b * 32 + (lane & 31)] += acc[a][b][r];
        if ((lane & 31) == 0)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[CB_K * CB_K + a * 32 + cb_row(r, hi)] += sum[a][r];
      }
    }
    __syncthreads();
  }
  float* out = partial + (long)blockIdx.x * CB_GRAM;
  for (int i = threadIdx.x; i < CB_GRAM; i += 256) out[i] = red[i];
}
// out[i] = sum over nb partial vectors of partial[b * n + i], fp64 sums: 32 entries x 8 slices of the partials per 256 threads
template <typename O>
__global__ void __launch_bounds__(256) conv1x1_partials_reduce_k(const float* __restrict__ partial, int nb, int n, O* __restrict__ out) {
  __shared__ double sm[8][32];
  const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + j;
  double a = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int b = sl; b < nb; b += 8) a += (double)partial[(long)b * n + i];
  }
  sm[sl][j] = a;
  __syncthreads();
  if (sl == 0 && i < n) {
#pragma unroll
    for (int q = 1; q < 8; ++q) a += sm[q][j];
    out[i] = (O)a;
  }
}

// t_i = sum_j S[i][j] w[j] for lane i (S in LDS, stride 65 doubles; w in LDS), wave-wide helpers
__device__ __forceinline__ double cb_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void cb_load_gram(const double* __restrict__ gram, double* Ss, double* ss) {
  for (int i = threadIdx.x; i < CB_K * CB_K; i += blockDim.x) Ss[(i >> 6) * 65 + (i & 63)] = gram[i];
  for (int i = threadIdx.x; i < CB_K; i += blockDim.x) ss[i] = gram[CB_K * CB_K + i];
}

// ------------------------------------------------------------------------------------------------ forward coefficients
// one wave per output channel (4 per workgroup): mean, variance from the moments; coef = [a | b]: y = act(a z + b)
__global__ void __launch_bounds__(256) conv1x1_bn_finalize_k(const double* __restrict__ gram, const bf16_t* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ coef,
                                                             int Cout, double n, float eps, float momentum) {
  __shared__ double Ss[CB_K * 65];
  __shared__ double ss[CB_K];
  __shared__ double ws[4][CB_K];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  cb_load_gram(gram, Ss, ss);
  const int c = blockIdx.x * 4 + wv;
  const double wi = c < Cout ? (double)bf2f(w[(long)c * CB_K + lane]) : 0.0;
  ws[wv][lane] = wi;
  __syncthreads();
  double t = 0.0;
#pragma unroll 8
  for (int j = 0; j < CB_K; ++j) t += Ss[lane * 65 + j] * ws[wv][j];
  const double mean = cb_wave_sum(wi * ss[lane]) / n;
  const double ez2 = cb_wave_sum(wi * t) / n;
  if (c >= Cout || lane != 0) return;
  double var = ez2 - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_rstd[c] = rstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
  const float a = gamma[c] * rstd;
  coef[c] = a;
  coef[Cout + c] = __fmaf_rn(-(float)mean, a, beta[c]);
}

// ------------------------------------------------------------------------------------------------ fused forward
// workgroup = 4 waves; tile = 32 tokens; wave w owns output channels [128 (4 blockIdx.y + w), +128): its 128 x 64 weights stay in registers.
// XCD x (blockIdx.x % 8) walks the position tiles p = x (mod 8), image fastest: the images of one position tile share the rows of `pos` in ONE L2.
#ifndef CB_EPI_GROUP
#define CB_EPI_GROUP 4
#endif
typedef unsigned cb_u32x4 __attribute__((ext_vector_type(4)));
#ifdef CB_NT_STORE
#define CB_STORE16(P_, V_) { const uint4 v__ = (V_); __builtin_nontemporal_store((cb_u32x4){v__.x, v__.y, v__.z, v__.w}, (cb_u32x4*)(P_)); }
#else
#define CB_STORE16(P_, V_) *(uint4*)(P_) = (V_)
#endif
#define CB_YLD 136                      // LDS row stride (bf16) of the wave's [32 tokens][128 channels] output tile
template <bool HAS_POS>
__global__ void __launch_bounds__(256) conv1x1_bn_act_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ coef,
                                                        const float* __restrict__ pos, bf16_t* __restrict__ y, bf16_t* __restrict__ q,
                                                        long HW, int B, int Cout, float slope) {
  __shared__ __attribute__((aligned(16))) bf16_t ytile_all[4][32 * CB_YLD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hi = lane >> 5, n = lane & 31;
  const int cg = blockIdx.y * 4 + wv;
  if (cg * 128 >= Cout) return;                                       // waves are independent: no workgroup barrier below
  bf16_t* ytile = ytile_all[wv];
  const int c0 = cg * 128;
  cb_bf16x8 Wf[4][4];                                                  // [channel block][K step]: B operand, lane (channel n, k group hi)
  float ca[4], cbv[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) Wf[jb][s] = *(const cb_bf16x8*)(w + (long)(c0 + jb * 32 + n) * CB_K + s * 16 + hi * 8);
    ca[jb] = coef[c0 + jb * 32 + n];
    cbv[jb] = coef[Cout + c0 + jb * 32 + n];
  }
  const int xcd = blockIdx.x & 7, wgx = gridDim.x >> 3, j0 = blockIdx.x >> 3;
  const long npt = (HW + 31) / 32;                                    // position tiles
  const long npx = npt > xcd ? (npt - xcd + 7) / 8 : 0;               // ... of this XCD
  const long nq = npx * B;
  cb_bf16x8 A[4];
  // the rows of tile t (clamped: rows past the end are computed and not stored); the loads of tile t + wgx are issued right after the MFMAs of tile t
#define CB_LOAD_A(T_)                                                                           \
  {                                                                                             \
    const long pl_ = (T_) / B;                                                                  \
    const int b_ = (int)((T_) - pl_ * B);                                                       \
    const long p_ = (pl_ * 8 + xcd) * 32 + n;                                                   \
    const bf16_t* xr = x + ((long)b_ * HW + (p_ < HW ? p_ : HW - 1)) * CB_K + hi * 8;           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) A[s] = *(const cb_bf16x8*)(xr + s * 16);      \
  }
  if (j0 < nq) CB_LOAD_A((long)j0)
  for (long t = j0; t < nq; t += wgx) {
    const long pl = t / B;
    const int b = (int)(t - pl * B);
    const long p0 = (pl * 8 + xcd) * 32;                              // first position of the tile
    cb_f32x16 acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[jb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], Wf[jb][s], acc[jb], 0, 0, 0);
    }
    if (t + wgx < nq) CB_LOAD_A(t + wgx)
    __builtin_amdgcn_wave_barrier();                                  // the previous tile's row reads are done (LDS is in order per wave)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z = __fmaf_rn(acc[jb][r], ca[jb], cbv[jb]);
        ytile[cb_row(r, hi) * CB_YLD + jb * 32 + n] = cb_bf(z > 0.f ? z : z * slope);
      }
    __builtin_amdgcn_wave_barrier();
    // rows back out of LDS as 16-byte pieces (16 lanes = one token's 256 bytes); CB_EPI_GROUP pieces per lane are handled together so that
    // their position loads are in flight at the same time
#pragma unroll
    for (int i0 = 0; i0 < 8; i0 += CB_EPI_GROUP) {
      float4 e0[CB_EPI_GROUP], e1[CB_EPI_GROUP];
      uint4 v[CB_EPI_GROUP];
      if (HAS_POS) {
#pragma unroll
        for (int i = 0; i < CB_EPI_GROUP; ++i) {
          const int idx = (i0 + i) * 64 + lane, tok = idx >> 4, piece = idx & 15;
          const long pr = p0 + tok < HW ? p0 + tok : HW - 1;
          const float* pp = pos + pr * Cout + c0 + piece * 8;
          e0[i] = *(const float4*)pp;
          e1[i] = *(const float4*)(pp + 4);
        }
      }
#pragma unroll
      for (int i = 0; i < CB_EPI_GROUP; ++i) {
        const int idx = (i0 + i) * 64 + lane, tok = idx >> 4, piece = idx & 15;
        v[i] = *(const uint4*)(ytile + tok * CB_YLD + piece * 8);
        if (p0 + tok < HW) CB_STORE16(y + ((long)b * HW + p0 + tok) * Cout + c0 + piece * 8, v[i]);
      }
      if (HAS_POS) {
#pragma unroll
        for (int i = 0; i < CB_EPI_GROUP; ++i) {
          const int idx = (i0 + i) * 64 + lane, tok = idx >> 4, piece = idx & 15;
          const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
          const float e[8] = {e0[i].x, e0[i].y, e0[i].z, e0[i].w, e1[i].x, e1[i].y, e1[i].z, e1[i].w};
          uint32_t o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(u[k] << 16) + e[2 * k], hi2 = __uint_as_float(u[k] & 0xffff0000u) + e[2 * k + 1];
            o[k] = cb_pk(lo, hi2);
          }
          if (p0 + tok < HW) CB_STORE16(q + ((long)b * HW + p0 + tok) * Cout + c0 + piece * 8, make_uint4(o[0], o[1], o[2], o[3]));
        }
      }
    }
  }
#undef CB_LOAD_A
}

// ------------------------------------------------------------------------------------------------ backward: masked gradient + column sums
// g = (dy1 + dy2) * act'(y) (either gradient may be absent; each has its own row stride), bf16; partial[block][C] = column sums of g
template <bool ONE, bool TWO>
__global__ void __launch_bounds__(256) conv1x1_bn_mask_k(const bf16_t* __restrict__ dy1, long ld1, const bf16_t* __restrict__ dy2, long ld2,
                                                         const bf16_t* __restrict__ y, bf16_t* __restrict__ g, float* __restrict__ partial, int C, long R,
                                                         int lpr, int rpi, float slope) {
  __shared__ float sm[256][8];
  const int t = threadIdx.x, ln = t % lpr, rs = t / lpr;
  const int c0 = ln * 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (rs < rpi) {
    const long step = (long)gridDim.x * rpi;
    for (long r = (long)blockIdx.x * rpi + rs; r < R; r += 2 * step) {                  // two rows per trip: six 16-byte loads in flight per lane
      const long r1 = r + step;
      const bool two = r1 < R;
      float a0[8], b0[8], y0[8], a1[8], b1[8], y1[8];
      V8<bf16_t>::ld(y + r * C + c0, y0);
      if (ONE) V8<bf16_t>::ld(dy1 + r * ld1 + c0, a0);
      if (TWO) V8<bf16_t>::ld(dy2 + r * ld2 + c0, b0);
      if (two) {
        V8<bf16_t>::ld(y + r1 * C + c0, y1);
        if (ONE) V8<bf16_t>::ld(dy1 + r1 * ld1 + c0, a1);
        if (TWO) V8<bf16_t>::ld(dy2 + r1 * ld2 + c0, b1);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gg = (ONE ? a0[k] : 0.f) + (TWO ? b0[k] : 0.f);
        gg = y0[k] > 0.f ? gg : gg * slope;
        gg = bf2f(f2bf(gg));                                            // the sums are those of the values the GEMMs will read
        a0[k] = gg;
        acc[k] += gg;
      }
      V8<bf16_t>::st(g + r * C + c0, a0);
      if (two) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float gg = (ONE ? a1[k] : 0.f) + (TWO ? b1[k] : 0.f);
          gg = y1[k] > 0.f ? gg : gg * slope;
          gg = bf2f(f2bf(gg));
          a1[k] = gg;
          acc[k] += gg;
        }
        V8<bf16_t>::st(g + r1 * C + c0, a1);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[t][k] = rs < rpi ? acc[k] : 0.f;
  __syncthreads();
  if (t < lpr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
      for (int q2 = 0; q2 < rpi; ++q2) s += sm[t + q2 * lpr][k];
      partial[(long)blockIdx.x * C + c0 + k] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: the small-matrix algebra
// one wave per output channel c (lane i = input channel): d_gamma, d_beta, dW[c][:], column c of Wd = a_c w_c (bf16), and the per-channel
// factors of the rank-64 corrections: k2[c] = rstd a m2, k0[c] = (mean rstd m2 - m1) a
__global__ void __launch_bounds__(256) conv1x1_bn_bwd_finalize_k(const float* __restrict__ GT, const float* __restrict__ m1s, const double* __restrict__ gram,
                                                                 const bf16_t* __restrict__ w, const float* __restrict__ gamma,
                                                                 const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dW,
                                                                 bf16_t* __restrict__ Wd, float* __restrict__ k2, float* __restrict__ k0, int Cout, double n) {
  __shared__ double Ss[CB_K * 65];
  __shared__ double ss[CB_K];
  __shared__ double ws[4][CB_K];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  cb_load_gram(gram, Ss, ss);
  const int c = blockIdx.x * 4 + wv;
  const bool ok = c < Cout;
  const double wi = ok ? (double)bf2f(w[(long)c * CB_K + lane]) : 0.0;
  ws[wv][lane] = wi;
  __syncthreads();
  if (!ok) return;
  double t = 0.0;
#pragma unroll 8
  for (int j = 0; j < CB_K; ++j) t += Ss[lane * 65 + j] * ws[wv][j];
  const double mean = (double)save_mean[c], rstd = (double)save_rstd[c], a = (double)gamma[c] * rstd;
  const double m1 = (double)m1s[c], Gi = (double)GT[(long)c * CB_K + lane], si = ss[lane];
  const double dg = rstd * cb_wave_sum(wi * (Gi - m1 * si / n));
  const double m1n = m1 / n, m2n = dg / n;
  const double ui = rstd * (t - si * mean);
  dW[(long)c * CB_K + lane] = (float)(a * (Gi - si * m1n - ui * m2n));
  Wd[(long)lane * (Cout + CB_K) + c] = f2bf((float)(a * wi));                   // A1^T: row = input channel, column = output channel
  if (lane == 0) {
    dgamma[c] = (float)dg;
    dbeta[c] = (float)m1;
    k2[c] = (float)(rstd * a * m2n);
    k0[c] = (float)((mean * rstd * m2n - m1n) * a);
  }
}
// Wd[j][Cout + i] = A2[i][j] = - sum_c k2[c] w[c][i] w[c][j] (symmetric);  c0[j] = sum_c k0[c] w[c][j]  (fp32)
__global__ void __launch_bounds__(256) conv1x1_bn_bwd_a2_k(const bf16_t* __restrict__ w, const float* __restrict__ k2, const float* __restrict__ k0,
                                                           bf16_t* __restrict__ Wd, float* __restrict__ c0v, int Cout) {
  __shared__ float sm[4][CB_K];
  const int i = blockIdx.x, j = threadIdx.x & 63, sl = threadIdx.x >> 6;                 // four slices of the output channels per (i, j)
  float a = 0.f;
  if (i < CB_K) {
    for (int c = sl; c < Cout; c += 4) a -= k2[c] * bf2f(w[(long)c * CB_K + i]) * bf2f(w[(long)c * CB_K + j]);
  } else {
    for (int c = sl; c < Cout; c += 4) a += k0[c] * bf2f(w[(long)c * CB_K + j]);
  }
  sm[sl][j] = a;
  __syncthreads();
  if (sl == 0) {
    a = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
    if (i < CB_K) Wd[(long)j * (Cout + CB_K) + Cout + i] = f2bf(a);
    else c0v[j] = a;
  }
}

// ------------------------------------------------------------------------------------------------ backward: data gradient
// dX (rows, 64) = [g | x] Wd^T + c0 with Wd (64, Cout + 64) resident in LDS (75 KB at Cout = 512): one streaming pass over g (the 0.8 GB operand)
// and x.  Computed transposed, D[channel][token] = Wd-fragment x row-fragment: both fragments are 16 contiguous bytes (LDS / HBM), a lane ends
// up with 4 consecutive channels of one token per accumulator group -> 8-byte stores.  Wave = 32 tokens; the rows are fetched eight K-steps
// (8 x 16 B per lane) ahead of the MFMAs that use them, across tile boundaries.
__global__ void __launch_bounds__(256) conv1x1_bn_dgrad_k(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wd,
                                                          const float* __restrict__ c0, bf16_t* __restrict__ dx, long rows, int Cout) {
  extern __shared__ __attribute__((aligned(16))) bf16_t wl[];                             // [64][Cout + 64 + 8]
  const int KP = Cout + CB_K, LD = KP + 8, ppr = KP / 8;
  for (int idx = threadIdx.x; idx < CB_K * ppr; idx += 256) {
    const int r = idx / ppr, pc = idx - r * ppr;
    *(uint4*)(wl + r * LD + pc * 8) = *(const uint4*)(Wd + (long)r * KP + pc * 8);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hi = lane >> 5, n = lane & 31;
  float c0r[2][16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) c0r[mt][r] = c0[mt * 32 + cb_row(r, hi)];
  const bf16_t* wa = wl + n * LD + hi * 8;                                                // + mt * 32 * LD + k * 16
  const long ntile = (rows + 31) / 32, tstep = (long)gridDim.x * 4;
  const int nch = Cout / 128;                                                             // chunks of 8 K-steps over g
  cb_bf16x8 Fa[8], Fb[8], Fx[4];
#define CB_ROWPTR(T_) (((T_) * 32 + n < rows) ? (T_) * 32 + n : rows - 1)
#define CB_LOADG(F_, T_, C_) { const bf16_t* gr_ = g + CB_ROWPTR(T_) * Cout + (C_) * 128 + hi * 8; _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) F_[s_] = *(const cb_bf16x8*)(gr_ + s_ * 16); }
#define CB_MMA(F_, C_)                                                                                                         \
  _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                                                           \
    const cb_bf16x8 w0_ = *(const cb_bf16x8*)(wa + ((C_) * 8 + s_) * 16), w1_ = *(const cb_bf16x8*)(wa + 32 * LD + ((C_) * 8 + s_) * 16); \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0_, F_[s_], acc0, 0, 0, 0);                                                 \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1_, F_[s_], acc1, 0, 0, 0);                                                 \
  }
  long tile = (long)blockIdx.x * 4 + wv;
  if (tile < ntile) CB_LOADG(Fa, tile, 0)
  for (; tile < ntile; tile += tstep) {
    {
      const bf16_t* xr = x + CB_ROWPTR(tile) * CB_K + hi * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) Fx[s] = *(const cb_bf16x8*)(xr + s * 16);
    }
    cb_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) CB_LOADG(Fb, tile, c + 1)
      CB_MMA(Fa, c)
      if (c + 2 < nch) CB_LOADG(Fa, tile, c + 2)
      else if (tile + tstep < ntile) CB_LOADG(Fa, tile + tstep, 0)
      if (c + 1 < nch) CB_MMA(Fb, c + 1)
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const cb_bf16x8 w0 = *(const cb_bf16x8*)(wa + Cout + s * 16), w1 = *(const cb_bf16x8*)(wa + 32 * LD + Cout + s * 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, Fx[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, Fx[s], acc1, 0, 0, 0);
    }
    const long row = tile * 32 + n;
    if (row < rows) {
      bf16_t* o = dx + row * CB_K + 4 * hi;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 v0, v1;
        v0.x = cb_pk(acc0[4 * r4] + c0r[0][4 * r4], acc0[4 * r4 + 1] + c0r[0][4 * r4 + 1]);
        v0.y = cb_pk(acc0[4 * r4 + 2] + c0r[0][4 * r4 + 2], acc0[4 * r4 + 3] + c0r[0][4 * r4 + 3]);
        v1.x = cb_pk(acc1[4 * r4] + c0r[1][4 * r4], acc1[4 * r4 + 1] + c0r[1][4 * r4 + 1]);
        v1.y = cb_pk(acc1[4 * r4 + 2] + c0r[1][4 * r4 + 2], acc1[4 * r4 + 3] + c0r[1][4 * r4 + 3]);
        *(uint2*)(o + 8 * r4) = v0;
        *(uint2*)(o + 32 + 8 * r4) = v1;
      }
    }
  }
#undef CB_ROWPTR
#undef CB_LOADG
#undef CB_MMA
}

// ================================================================================================ C ABI
static inline bool cb_ok(int Cin, int Cout) { return Cin == CB_K && Cout > 0 && Cout % 128 == 0 && Cout <= 2048; }

// bytes of the scratch buffer of ge_conv1x1_bn_stats / ge_conv1x1_bn_bwd_mask (partial Gram matrices / partial column sums)
extern "C" size_t ge_conv1x1_bn_workspace(int Cin, int Cout) {
  if (!cb_ok(Cin, Cout)) return 0;
  const size_t a = (size_t)CB_GRAM_WG * CB_GRAM * sizeof(float), b = (size_t)1024 * Cout * sizeof(float);
  return a > b ? a : b;
}

// Moments of the input and the BatchNorm coefficients of the 1x1 convolution output.  gram: CB_GRAM doubles (kept for the backward);
// coef: [a | b], 2 Cout floats; running statistics updated like F.batch_norm (may be NULL).
extern "C" int ge_conv1x1_bn_stats(const void* x, long rows, int Cin, const void* w, int Cout, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float eps, float momentum, double* gram, float* save_mean,
                                   float* save_rstd, float* coef, void* workspace, void* stream) {
  if (!x || !w || !gamma || !beta || !gram || !save_mean || !save_rstd || !coef || !workspace || rows <= 0) return GE_ERR_BAD_ARG;
  if (!cb_ok(Cin, Cout) || (((uintptr_t)x | (uintptr_t)w) & 15)) return GE_ERR_UNSUPPORTED;
  hipStream_t s = ge_stream(stream);
  const long nblk = (rows + 63) / 64;
  const int nb = (int)(nblk < 4L * CB_GRAM_WG ? (nblk + 3) / 4 : CB_GRAM_WG);
  conv1x1_gram_k<<<nb, 256, 0, s>>>((const bf16_t*)x, rows, (float*)workspace);
  GE_LAUNCH_CHECK();
  conv1x1_partials_reduce_k<double><<<(CB_GRAM + 31) / 32, 256, 0, s>>>((const float*)workspace, nb, CB_GRAM, gram);
  GE_LAUNCH_CHECK();
  conv1x1_bn_finalize_k<<<(Cout + 3) / 4, 256, 0, s>>>(gram, (const bf16_t*)w, gamma, beta, save_mean, save_rstd, running_mean, running_var, coef,
                                                       Cout, (double)rows, eps, momentum);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// y[b, p, :] = act(coef_a * (x[b, p, :] W^T) + coef_b) (bf16, rows = B * HW of Cout channels), q = y + pos[p, :] (pos fp32 (HW, Cout); q, pos may be NULL)
extern "C" int ge_conv1x1_bn_act_fwd(const void* x, const void* w, const float* coef, const float* pos, void* y, void* q, int B, long HW, int Cin,
                                     int Cout, float slope, void* stream) {
  if (!x || !w || !coef || !y || B <= 0 || HW <= 0 || (!pos) != (!q)) return GE_ERR_BAD_ARG;
  if (!cb_ok(Cin, Cout) || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)q | (uintptr_t)pos) & 15)) return GE_ERR_UNSUPPORTED;
  const int cus = ge_cu_count();
  if (!cus) return GE_ERR_BAD_ARG;
  hipStream_t s = ge_stream(stream);
  const long ntile = ((HW + 31) / 32) * B;
#ifndef CB_WGS_PER_CU
#define CB_WGS_PER_CU 4
#endif
  long gx = (long)cus * CB_WGS_PER_CU;                                            // four workgroups per CU (35 KB of LDS, <= 128 registers); a multiple of 8: whole XCD rounds
  if (gx > ntile) gx = ntile;
  gx = (gx + 7) / 8 * 8;
  const dim3 grid((unsigned)gx, (unsigned)((Cout / 128 + 3) / 4));
  if (pos) conv1x1_bn_act_k<true><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)w, coef, pos, (bf16_t*)y, (bf16_t*)q, HW, B, Cout, slope);
  else conv1x1_bn_act_k<false><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)w, coef, nullptr, (bf16_t*)y, nullptr, HW, B, Cout, slope);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// g (rows x C, bf16) = (dy1 + dy2) * act'(y), colsum (C floats) = its column sums.  dy1 / dy2: bf16 rows of C channels with row strides ld1 / ld2
// (elements; a gradient that arrives as a channel slice of a wider map is read in place); either may be NULL, not both.
extern "C" int ge_conv1x1_bn_bwd_mask(const void* dy1, long ld1, const void* dy2, long ld2, const void* y, void* g, float* colsum, void* workspace,
                                      long rows, int C, float slope, void* stream) {
  if ((!dy1 && !dy2) || !y || !g || !colsum || !workspace || rows <= 0 || C <= 0) return GE_ERR_BAD_ARG;
  if (C % 8 || C > 2048 || (dy1 && ld1 % 8) || (dy2 && ld2 % 8) || (((uintptr_t)dy1 | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)g) & 15)) return GE_ERR_UNSUPPORTED;
  hipStream_t s = ge_stream(stream);
  const int lpr = C / 8;
  if (lpr > 256) return GE_ERR_UNSUPPORTED;
  const int rpi = 256 / lpr;
  long nb = (rows + (long)rpi * 8 - 1) / ((long)rpi * 8);
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  float* part = (float*)workspace;
#define CB_MASK(A_, B_) conv1x1_bn_mask_k<A_, B_><<<(unsigned)nb, 256, 0, s>>>((const bf16_t*)dy1, ld1, (const bf16_t*)dy2, ld2, (const bf16_t*)y, (bf16_t*)g, part, C, rows, lpr, rpi, slope)
  if (dy1 && dy2) CB_MASK(true, true);
  else if (dy1) CB_MASK(true, false);
  else CB_MASK(false, true);
#undef CB_MASK
  GE_LAUNCH_CHECK();
  conv1x1_partials_reduce_k<float><<<(C + 31) / 32, 256, 0, s>>>(part, (int)nb, C, colsum);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// The algebra after G = g^T X (GT: (Cout, 64) fp32, ge_conv1x1_nhwc_wgrad) and colsum: d_gamma, d_beta, dW (Cout, 64) fp32, and the operands of
// dX = [g | x] Wd^T + c0 (ge_conv1x1_bn_dgrad): Wd (64, Cout + 64) bf16, c0 (64) fp32.  scratch: 2 Cout floats.
extern "C" int ge_conv1x1_bn_bwd_finalize(const float* GT, const float* colsum, const double* gram, const void* w, const float* gamma,
                                          const float* save_mean, const float* save_rstd, long rows, int Cin, int Cout, float* dgamma, float* dbeta,
                                          float* dW, void* Wd, float* c0, float* scratch, void* stream) {
  if (!GT || !colsum || !gram || !w || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !dW || !Wd || !c0 || !scratch || rows <= 0) return GE_ERR_BAD_ARG;
  if (!cb_ok(Cin, Cout)) return GE_ERR_UNSUPPORTED;
  hipStream_t s = ge_stream(stream);
  conv1x1_bn_bwd_finalize_k<<<(Cout + 3) / 4, 256, 0, s>>>(GT, colsum, gram, (const bf16_t*)w, gamma, save_mean, save_rstd, dgamma, dbeta, dW, (bf16_t*)Wd,
                                                           scratch, scratch + Cout, Cout, (double)rows);
  GE_LAUNCH_CHECK();
  conv1x1_bn_bwd_a2_k<<<CB_K + 1, 256, 0, s>>>((const bf16_t*)w, scratch, scratch + Cout, (bf16_t*)Wd, c0, Cout);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// dx (rows, 64) bf16 = g (rows, Cout) Wd[:, :Cout]^T + x (rows, 64) Wd[:, Cout:]^T + c0: the convolution's data gradient with the BatchNorm scale
// folded into the weights plus the rank-64 mean-subtraction terms, one pass.  Cout <= 1024 (Wd lives in LDS).
extern "C" int ge_conv1x1_bn_dgrad(const void* g, const void* x, const void* Wd, const float* c0, void* dx, long rows, int Cin, int Cout, void* stream) {
  if (!g || !x || !Wd || !c0 || !dx || rows <= 0) return GE_ERR_BAD_ARG;
  if (!cb_ok(Cin, Cout) || Cout > 1024 || (((uintptr_t)g | (uintptr_t)x | (uintptr_t)Wd | (uintptr_t)dx) & 15)) return GE_ERR_UNSUPPORTED;
  const int cus = ge_cu_count();
  if (!cus) return GE_ERR_BAD_ARG;
  const size_t smem = (size_t)CB_K * (Cout + CB_K + 8) * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv1x1_bn_dgrad_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GE_ERR_UNSUPPORTED;
    attr_set = true;
  }
  const long ntile = (rows + 31) / 32;
  long wgs = (long)cus * (smem <= 78 * 1024 ? 2 : 1);
  if (wgs > (ntile + 3) / 4) wgs = (ntile + 3) / 4;
  conv1x1_bn_dgrad_k<<<(unsigned)wgs, 256, smem, ge_stream(stream)>>>((const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)Wd, c0, (bf16_t*)dx, rows, Cout);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
