// Shared definitions of the deformable-attention kernels (msda.hip: streaming / binned kernels, msda_win.hip: the
// LDS-window kernels).
#pragma once
#include "common.h"

#define MSDA_MAX_L 8
struct MsdaLevels { int H[MSDA_MAX_L]; int W[MSDA_MAX_L]; int start[MSDA_MAX_L]; };


// CPL channels per lane: 16-byte accesses for both storage types (fp32: 4 channels, bf16: 8 channels)
template <typename T> struct Lanes;
template <> struct Lanes<float> { static constexpr int CPL = 4; };
template <> struct Lanes<bf16_t> { static constexpr int CPL = 8; };
template <typename T> struct VecL;
template <> struct VecL<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[4]) { V8<float>::ld(p, v); }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { V8<float>::st(p, v); }
};
template <> struct VecL<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[8]) {
    const uint4 t = *(const uint4*)p;
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
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


// Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MB L2.  The sampling kernels read a sliding
// neighbourhood of `value`, so neighbouring queries should meet in the SAME L2: XCD x works on the x-th contiguous eighth of
// the (batch, query) range (one image per XCD at 8 images) instead of every eighth workgroup of all of it.
// rocprofv3 FETCH_SIZE, forward kernel, 8x352x1120: 10.0 GB per launch with the plain mapping.
#define MSDA_XCDS 8
__device__ __forceinline__ long msda_xcd_block(unsigned bid, unsigned nblk) {      // nblk is a multiple of MSDA_XCDS
  return (long)(bid % MSDA_XCDS) * (nblk / MSDA_XCDS) + bid / MSDA_XCDS;
}
static inline unsigned msda_grid(long n_items, int per_block) {
  long b = (n_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  b = (b + MSDA_XCDS - 1) / MSDA_XCDS * MSDA_XCDS;
  return (unsigned)b;
}


// <gradient row piece, value row piece> over the 16 bytes a lane holds: 4 fp32 FMAs, or 4 x v_dot2c_f32_bf16 on the raw
// bf16 pairs (no bf16 -> f32 unpacking: 16 instead of 64 VALU per sampling point for the four corners)
typedef unsigned int lw_raw_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <typename T> struct RowDot;
template <> struct RowDot<float> {
  static __device__ __forceinline__ float dot(const lw_raw_t& a, const lw_raw_t& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += __uint_as_float(a[i]) * __uint_as_float(b[i]);
    return s;
  }
};
template <> struct RowDot<bf16_t> {
  static __device__ __forceinline__ float dot(const lw_raw_t& a, const lw_raw_t& b) {
    // element copies first: hipcc (ROCm 7.2) miscompiles __builtin_bit_cast(bf16x2_t, a[i]) on a vector-element lvalue — every i reads
    // element 0 (one dword loaded, dotted four times; tests/test_kernels_gpu.py::test_msda_bf16_gradients_vs_oracle pins this)
    const unsigned int a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    float s = 0.f;
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a0), __builtin_bit_cast(bf16x2_t, b0), s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a1), __builtin_bit_cast(bf16x2_t, b1), s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a2), __builtin_bit_cast(bf16x2_t, b2), s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a3), __builtin_bit_cast(bf16x2_t, b3), s, false);
    return s;
  }
};


// Row indices and pitches are formed with the 24-bit integer multiply (v_mul_i32_i24 / v_mad_i32_i24: full rate on CDNA, where the
// 32-bit v_mul_lo_u32 takes four passes — six of those per sampling point were ~15 % of the d_loc / d_attw kernel).  Operands are
// map coordinates, widths and row pitches; msda_levels() refuses maps of 2^23 positions or more.
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

static inline int msda_levels(const int* spatial_hw, int L, int Nv, MsdaLevels& lv) {
  if (L < 1 || L > MSDA_MAX_L) return GE_ERR_UNSUPPORTED;
  long start = 0;
  for (int l = 0; l < L; ++l) {
    lv.H[l] = spatial_hw[2 * l]; lv.W[l] = spatial_hw[2 * l + 1]; lv.start[l] = (int)start;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return GE_ERR_BAD_ARG;
    if ((long)lv.H[l] * lv.W[l] >= (1L << 23)) return GE_ERR_UNSUPPORTED;          // mul24 index arithmetic
    start += (long)lv.H[l] * lv.W[l];
  }
  return start == Nv ? GE_OK : GE_ERR_BAD_ARG;
}


// ---- d_value by binning (msda.hip: count / scan / fill / drain; msda_drain_mfma.hip: the bf16 MFMA drain)
#define MSDA_TW 8
#define MSDA_TH 4
#define MSDA_TILE 32           // positions per bin = one 32-register accumulator block per lane
#define MSDA_CHUNK 4096
typedef float f32x32_t __attribute__((ext_vector_type(32)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct MsdaBins { int first_tile[MSDA_MAX_L + 1]; int ntx[MSDA_MAX_L]; };   // tiles of level l: [first_tile[l], first_tile[l+1]), ntx per row

struct MsdaWs {            // device workspace carved by the host wrapper
  int* cnt; int* seg_hist; int* chunk_first; long* offset; int* ctrl; int4* entries;   // ctrl[0] = total chunks, ctrl[2+x] = next chunk of XCD x's share
  int* order;              // drain work order: item -> chunk id (msda_order_k), grouped by (image, head, query range of the chunk's first record)
};

// bf16 drain of the binned d_value on the matrix cores (msda_drain_mfma.hip); tr = use ds_read_b64_tr_b16 for the B operand
// rec8 = 8-byte records {query << 7 | corner, bf16 weight | 8-bit frac x << 16 | 8-bit frac y << 24} (msda_hist_raw_k)
int msda_drain_mfma_launch(const MsdaLevels& lv, const MsdaBins& bins, const MsdaWs& ws, const void* gout, float* d_value, int nbins,
                           int Nv, int Nq, int nH, int L, bool tr, hipStream_t s, bool rec8 = false);


// LDS-window kernels (msda_win.hip); query geometry = n_qseg (H, W) segments of queries in raster order
int msda_win_supported(int B, int Nq, int nH, int L, int P, int Nv);
int msda_fwd_win_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const float* loc, const float* attw,
                        void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, bool pre, hipStream_t s);
// raw projection inputs of the fused prepare + forward (msda_win.hip)
struct MwRaw {
  const void* off; long off_ld; const void* logit; long logit_ld;          // (B*Nq, ld) rows; columns (head, level, point[, xy])
  const float* ref; long ref_sb, ref_sq, ref_sl;                             // reference points (B, Nq, L, 2), strides in elements
  float* loc_out; float* attw_out;
};
int msda_fwd_win_raw_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const MwRaw& rw, void* out, int B,
                            int Nv, int Nq, int nH, int L, int P, int dtype, hipStream_t s);
int msda_bwd_lw_win_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const float* loc,
                           const float* attw, const void* gout, float* d_loc, float* d_attw, int B, int Nv, int Nq, int nH, int L,
                           int P, int dtype, bool pre, hipStream_t s);
