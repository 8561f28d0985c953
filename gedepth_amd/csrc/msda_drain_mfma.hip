// d_value of the deformable attention (mmcv ms_deform_attn backward; reference call sites depth/models/necks/hahi.py:279-289,
// 316-325) — second half of the binned scatter, bf16 storage, on the matrix cores.
//
// After binning (msda.hip: count / scan / fill) a chunk is a list of (point, tile) records {query, corner of the sampling
// point inside the 8 x 4 value tile, weight, frac x, frac y}.  Summing a chunk into its tile IS a contraction:
//
//     dV[pos, ch] = sum_rec C[pos, rec] * G[rec, ch]        pos: 32 tile positions, ch: 64 channels of the head
//
// with G[rec, :] the gradient row of the record's (query, head) and C the bilinear coefficients (<= 4 non-zeros per
// column).  The VALU drain (msda_drain_k) walks the records one by one, 9 wave-instructions each, and is bound by its
// own instruction stream.  Here one v_mfma_f32_32x32x16_bf16 takes 16 records x 32 channels:
//   * C is never formed per position: one lane per record computes its four coefficients and drops them as bf16 into a
//     zeroed LDS image that already has the A-operand layout ([k-group][position][8 records]) — 4 ds_write_b16 per record
//     (and 4 to clear them again), then ONE ds_read_b128 per lane and K step;
//   * G rows are gathered 8 per instruction (16 B per lane, as before) into an LDS stage [channel half][record][32] and
//     reach the B-operand layout (8 records of one channel per lane) through gfx950's transposing LDS read
//     ds_read_b64_tr_b16: 2 reads per operand;
//   * no sort by corner class, no per-record scalar work: ~1.5 wave-instructions per record, 8 MFMA per 64 records.
// Coefficients are rounded to bf16 (2^-9 relative, the rounding d_value gets anyway when it is returned in the storage
// type); products are exact, sums fp32.  The exact-fp32 parity path keeps the VALU drain.
#include "msda.h"

typedef __bf16 md_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 md_bf16x4 __attribute__((ext_vector_type(4)));
typedef float md_f32x16 __attribute__((ext_vector_type(16)));

#define MD_BLK 64                      // records per block = 4 K steps
#define MD_HALF (MD_BLK * 32 + 32)     // bf16 elements of one channel-half image of the stage: [64 records][32 channels] + 64 B skew
#define MD_STAGE (2 * MD_HALF)
#define MD_AIMG (MD_BLK * 32)          // [4 K steps][2 k-groups][32 positions][8 records]
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
__device__ __forceinline__ bf16_t md_bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }   // v_cvt_pk_bf16_f32, RNE

// REC8: 8-byte records (bf16 weight, 8-bit fractions: the coefficients are rounded to bf16 in the image anyway; a fraction is taken at
// the centre of its 1/256 step, so a coefficient moves by <= 2^-9 of the point's weight) — half the record bytes of fill and drain.
template <bool REC8> struct MdRec;
// Records are read exactly once: loaded with the non-temporal hint so that the 2.3 GB stream does not push the gradient rows (which the
// range-grouped work order keeps re-using, msda_order_k) out of the XCD's L2.
typedef int md_i32x4 __attribute__((ext_vector_type(4)));
typedef int md_i32x2 __attribute__((ext_vector_type(2)));
template <> struct MdRec<false> {
  typedef int4 T;
  static __device__ __forceinline__ T load(const T* p) { const md_i32x4 v = __builtin_nontemporal_load((const md_i32x4*)p); return make_int4(v.x, v.y, v.z, v.w); }
  static __device__ __forceinline__ T pad(int key) { return make_int4(key, 0, 0, 0); }
  static __device__ __forceinline__ int key(const T& e) { return e.x; }
  static __device__ __forceinline__ void coef(const T& e, float& w, float& ax, float& ay) { w = __int_as_float(e.y); ax = __int_as_float(e.z); ay = __int_as_float(e.w); }
};
template <> struct MdRec<true> {
  typedef int2 T;
  static __device__ __forceinline__ T load(const T* p) { const md_i32x2 v = __builtin_nontemporal_load((const md_i32x2*)p); return make_int2(v.x, v.y); }
  static __device__ __forceinline__ T pad(int key) { return make_int2(key, 0); }
  static __device__ __forceinline__ int key(const T& e) { return e.x; }
  static __device__ __forceinline__ void coef(const T& e, float& w, float& ax, float& ay) {
    w = __uint_as_float((uint32_t)e.y << 16);
    ax = ((float)(((uint32_t)e.y >> 16) & 255u) + 0.5f) * (1.f / 256.f);
    ay = ((float)((uint32_t)e.y >> 24) + 0.5f) * (1.f / 256.f);
  }
};

template <bool TR, bool REC8>
__global__ void __launch_bounds__(256) msda_drain_mfma_k(MsdaLevels lv, MsdaBins bins, MsdaWs ws, const bf16_t* __restrict__ gout,
                                                         float* __restrict__ d_value, int nbins, int Nv, int Nq, int nH, int L) {
  __shared__ __attribute__((aligned(16))) bf16_t stage_all[4 * MD_STAGE];
  __shared__ __attribute__((aligned(16))) bf16_t aimg_all[4 * MD_AIMG];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bf16_t* stage = stage_all + wv * MD_STAGE;
  bf16_t* aimg = aimg_all + wv * MD_AIMG;
  {                                                         // the coefficient image starts (and is kept) all-zero
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MD_AIMG / (64 * 8); ++i) *(uint4*)(aimg + (i * 64 + lane) * 8) = z;
  }
  const int ntiles = bins.first_tile[L];
  const int total = ws.ctrl[0];
  const int part0 = blockIdx.x % MSDA_XCDS;
  const int sub8 = lane & 7, row8 = lane >> 3;
  // stage slot of this lane's 16-byte piece of a gathered row: channels sub8*8 .. +7 -> half sub8/4, column (sub8%4)*8
  const int park_off = (sub8 >> 2) * MD_HALF + (sub8 & 3) * 8;
  // A-image slot of (record = lane, position m): ((kstep*2 + kg)*32 + m)*8 + k8 with kstep = lane/16, kg = (lane/8)&1, k8 = lane&7
  const int a_rec = ((lane >> 4) * 2 + ((lane >> 3) & 1)) * 256 + (lane & 7);
  // B operand: lane (n = lane&31, kg = lane>>5).  tr read t of K step ks fetches records ks*16 + kg*8 + 4t + (0..3): within its
  // 16-lane group lane i supplies the 8-byte piece (row i/4, columns 4(i%4)..) of the [4 records][16 channels] block
  const int tr_row = (lane >> 5) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
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
    if (ws.order) item = ws.order[item];                  // work order -> chunk id (msda_order_k, msda.hip: chunks grouped by query range; mode bit 6)
    int lo = 0, hi = nbins;                               // largest bin with chunk_first[bin] <= item
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ws.chunk_first[mid] <= item) lo = mid; else hi = mid; }
    const int bin = lo;
    const int nchunk = ws.chunk_first[bin + 1] - ws.chunk_first[bin];
    const int chunk = item - ws.chunk_first[bin];
    const int cnt = ws.cnt[bin];
    const int e_lo = chunk * MSDA_CHUNK, e_hi = min(cnt, e_lo + MSDA_CHUNK);
    typedef typename MdRec<REC8>::T rec_t;
    const rec_t* ent = (const rec_t*)ws.entries + ws.offset[bin] + e_lo;
    const int n = e_hi - e_lo;
    const int bh = bin / ntiles, tile = bin - bh * ntiles;
    const int b = bh / nH, head = bh - b * nH;
    int l = 0;
    while (l + 1 < L && tile >= bins.first_tile[l + 1]) ++l;
    const int tl = tile - bins.first_tile[l];
    const int ty = tl / bins.ntx[l], tx = tl - ty * bins.ntx[l];
    const bf16_t* grow = gout + ((long)b * Nq * nH + head) * 64 + sub8 * 8;     // + q * nH * 64
    const int qpitch = nH * 64;

    md_f32x16 acc0 = 0.f, acc1 = 0.f;
    const int nb = (n + MD_BLK - 1) / MD_BLK;
    u32x4_t R[8];
    rec_t E0, E1, E2;
    // a record slot past the end of the chunk: zero weight, the query of the chunk's first record (a row that exists)
    const rec_t padrec = MdRec<REC8>::pad(MdRec<REC8>::key(ent[0]));
#define MD_LOAD(DST, BLK) { DST = padrec; if ((BLK) * MD_BLK + lane < n) DST = MdRec<REC8>::load(ent + (BLK) * MD_BLK + lane); }
#define MD_GATHER(EE)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
    const int q = __shfl(MdRec<REC8>::key(EE) >> 7, i * 8 + row8, 64);                                       \
    R[i] = *(const u32x4_t*)(grow + (long)q * qpitch);                                         \
  }
#define MD_PARK()                                                                              \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) *(u32x4_t*)(stage + park_off + (i * 8 + row8) * 32) = R[i];
    // the four coefficients of this lane's record -> the A image (ZERO = true: clear the same slots again)
#define MD_COEF(EE, ZERO)                                                                      \
  {                                                                                            \
    const int lx = (MdRec<REC8>::key(EE) & 15) - 1, ly = ((MdRec<REC8>::key(EE) >> 4) & 7) - 1; \
    float w, ax, ay;                                                                            \
    MdRec<REC8>::coef(EE, w, ax, ay);                                                           \
    const bool xl = lx >= 0, xr = lx < MSDA_TW - 1, yt = ly >= 0, yb = ly < MSDA_TH - 1;        \
    const float wl = 1.f - ax, wt = w * (1.f - ay), wb = w * ay;                                \
    bf16_t* a00 = aimg + a_rec + (ly * MSDA_TW + lx) * 8;                                       \
    if (yt && xl) a00[0] = (ZERO) ? (bf16_t)0 : md_bf(wt * wl);                                 \
    if (yt && xr) a00[8] = (ZERO) ? (bf16_t)0 : md_bf(wt * ax);                                 \
    if (yb && xl) a00[MSDA_TW * 8] = (ZERO) ? (bf16_t)0 : md_bf(wb * wl);                       \
    if (yb && xr) a00[MSDA_TW * 8 + 8] = (ZERO) ? (bf16_t)0 : md_bf(wb * ax);                   \
  }
    MD_LOAD(E0, 0)
    MD_LOAD(E1, 1)
    MD_GATHER(E0)
    __builtin_amdgcn_wave_barrier();
    MD_COEF(E0, false)
    MD_PARK()
    __builtin_amdgcn_wave_barrier();
    for (int blk = 0; blk < nb; ++blk) {
      const bool more = blk + 1 < nb;
      if (more) { MD_GATHER(E1) }                               // next block's rows in flight during the MFMA phase
      MD_LOAD(E2, blk + 2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const md_bf16x8 A = *(const md_bf16x8*)(aimg + ((ks * 2 + (lane >> 5)) * 32 + (lane & 31)) * 8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          md_bf16x8 Bv;
          if (TR) {
            const bf16_t* p = stage + half * MD_HALF + (ks * 16 + tr_row) * 32 + tr_col;
            const md_bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(md_bf16x4, p));
            const md_bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(md_bf16x4, p + 4 * 32));
            Bv = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          } else {
            const bf16_t* p = stage + half * MD_HALF + (ks * 16 + (lane >> 5) * 8) * 32 + (lane & 31);
            uint32_t wq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) wq[e] = (uint32_t)p[(2 * e) * 32] | ((uint32_t)p[(2 * e + 1) * 32] << 16);
            Bv = __builtin_bit_cast(md_bf16x8, make_uint4(wq[0], wq[1], wq[2], wq[3]));
          }
          if (half == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc0, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
      MD_COEF(E0, true)
      if (more) {
        MD_COEF(E1, false)
        MD_PARK()
      }
      __builtin_amdgcn_wave_barrier();
      E0 = E1; E1 = E2;
    }
#undef MD_LOAD
#undef MD_GATHER
#undef MD_PARK
#undef MD_COEF
    // C/D layout: column = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (tile position)
    const int Wl = lv.W[l], Hl = lv.H[l];
    float* dst = d_value + (((long)b * Nv + lv.start[l]) * nH + head) * 64 + (lane & 31);
    const long pstride = (long)nH * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int y = ty * MSDA_TH + m / MSDA_TW, x = tx * MSDA_TW + m % MSDA_TW;
      if (y < Hl && x < Wl) {
        float* p = dst + ((long)y * Wl + x) * pstride;
        if (nchunk == 1) { p[0] = acc0[r]; p[32] = acc1[r]; }
        else { atomicAdd(p, acc0[r]); atomicAdd(p + 32, acc1[r]); }
      }
    }
  }
}

int msda_drain_mfma_launch(const MsdaLevels& lv, const MsdaBins& bins, const MsdaWs& ws, const void* gout, float* d_value, int nbins,
                           int Nv, int Nq, int nH, int L, bool tr, hipStream_t s, bool rec8) {
  if (rec8) msda_drain_mfma_k<true, true><<<256 * 3, 256, 0, s>>>(lv, bins, ws, (const bf16_t*)gout, d_value, nbins, Nv, Nq, nH, L);
  else if (tr) msda_drain_mfma_k<true, false><<<256 * 3, 256, 0, s>>>(lv, bins, ws, (const bf16_t*)gout, d_value, nbins, Nv, Nq, nH, L);
  else msda_drain_mfma_k<false, false><<<256 * 3, 256, 0, s>>>(lv, bins, ws, (const bf16_t*)gout, d_value, nbins, Nv, Nq, nH, L);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
