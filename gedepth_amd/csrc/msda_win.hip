// Deformable-attention sampling with LDS-staged value windows (gfx950): forward and the d_loc / d_attw backward pass.
// Same op as msda_fwd_k / msda_bwd_lw_k in msda.hip (mmcv ms_deform_attn, reference call sites
// depth/models/necks/hahi.py:279-289,316-325), different decomposition.
//
// Why: the streaming kernels are bound by the number of vector-memory (TA) instructions, ~25 cycles each per CU whatever
// they fetch: per sampling point 4 row gathers + 2 broadcast loads of its location / weight.  Queries that
// are neighbours in the image sample neighbouring value rows (self-attention: reference point = own pixel centre;
// cross-attention: a smooth function of the position, hahi.py:294-302), so a TILE of queries touches a compact window of
// each level — each row of it hundreds of times.  Here a workgroup owns (image, head, 16x16 query tile) and, level by level:
//   A  one sampling point per lane: coalesced 8-byte / 4-byte loads of loc / attw (2 TA instructions per 8 points
//      instead of 16), tap coordinates, bounding box of the taps of the whole tile (wave min/max + 4 LDS atomics);
//   B  if the box has <= CAP rows it is copied into LDS once (128-byte rows of this head, 16 bytes per lane);
//   C  the 4 x 8 x 256 taps are ds_read_b128 reads of the window (LDS: 256 B/clk/CU instead of the 64 B/clk/CU vector
//      L1 path); point data reach the 8 lanes of a (query, head) group by ds_bpermute.  A box that does not fit (scattered
//      reference points after training, coarse-level queries sampling the finest level) takes the same code on global
//      pointers — a per-(tile, level) decision, so correctness never depends on locality.
// No MFMA: a gather, not a contraction.
#include "msda.h"
#include <limits.h>

#ifndef MW_BYTES
#define MW_BYTES 51200                    // LDS window: 400 bf16 rows / 200 fp32 rows of 64 channels; 3 workgroups per CU
#endif
#define MW_QPG 4                          // queries per lane group (accumulators live in registers across the level loop)

struct MsdaQGrid {                        // query tiling: segments of (H, W) queries in raster order (the levels for
  int nseg;                               // self-attention, one 176x560 map for the cross-attention; 1 x Nq if unknown)
  int H[MSDA_MAX_L], W[MSDA_MAX_L], start[MSDA_MAX_L];
  int tiles_x[MSDA_MAX_L], tile_first[MSDA_MAX_L + 1];
};

template <typename T> struct WinGeom {
  static constexpr int CPL = Lanes<T>::CPL;          // channels per lane (16 bytes)
  static constexpr int G = 64 / CPL;                 // lanes per (query, head) group = per value row
  static constexpr int NG = 256 / G;                 // groups per workgroup
  static constexpr int TQ = NG * MW_QPG;             // queries per tile: 256 (bf16), 128 (fp32)
  static constexpr int TQW = 16, TQH = TQ / 16;      // 16 x 16 / 8 x 16 (H x W)
  static constexpr int CAP = MW_BYTES / (64 * (int)sizeof(T));
};

// Tap geometry of one sampling point, branch-free: addresses clamped into the map, out-of-range corners carry zero
// weight (identical arithmetic to msda_fwd_k / msda_bwd_lw_k).
struct MwTap { int xa, xb, ya, yb; float ax, ay; bool kxa, kxb, kya, kyb; };
__device__ __forceinline__ MwTap mw_tap(float x, float y, int Wl, int Hl) {
  MwTap t;
  const float xc = fminf(fmaxf(x, -1.f), (float)Wl), yc = fminf(fmaxf(y, -1.f), (float)Hl);
  const float xf = floorf(xc), yf = floorf(yc);
  const int x0 = (int)xf, y0 = (int)yf;
  t.ax = xc - xf; t.ay = yc - yf;
  t.xa = min(max(x0, 0), Wl - 1); t.xb = min(max(x0 + 1, 0), Wl - 1);
  t.ya = min(max(y0, 0), Hl - 1); t.yb = min(max(y0 + 1, 0), Hl - 1);
  t.kxa = x0 >= 0; t.kxb = x0 + 1 < Wl; t.kya = y0 >= 0; t.kyb = y0 + 1 < Hl;
  return t;
}
__device__ __forceinline__ int mw_wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int mw_wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// RAW = true: the kernel consumes the raw outputs of the `sampling_offsets` / `attention_weights` linears (+ the reference
// points) instead of ready-made locations / weights, i.e. mmcv's view -> softmax over L*P -> `ref + off / (W_l, H_l)` happens in
// registers (same arithmetic as msda_prep_fwd_k), and WRITES loc / attw (fp32) for the backward kernels — the separate prepare
// pass (0.85 + 0.29 ms per step at the KITTI shape) and one read of loc / attw disappear; the stores ride on a VALU-bound kernel.
template <typename T> __device__ __forceinline__ void mw_ld_pair(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void mw_ld_pair<float>(const float* p, float& a, float& b) { const float2 t = *(const float2*)p; a = t.x; b = t.y; }
template <> __device__ __forceinline__ void mw_ld_pair<bf16_t>(const bf16_t* p, float& a, float& b) {
  const uint32_t t = *(const uint32_t*)p; a = __uint_as_float(t << 16); b = __uint_as_float(t & 0xffff0000u);
}
// Prepare pre-pass of the RAW kernel (launcher: L == 4, P == 8): softmax over the 32 logits of each of this lane group's (query,
// head) pairs and `ref + off / (W_l, H_l)` for its 32 points, written to loc_out / attw_out at once — lane sub < 8 owns point `sub` of
// the four levels, so a (query, head) leaves as four 64-byte + four 32-byte stores issued back to back (256 + 128 contiguous bytes:
// complete lines for the L2 to write back; storing level by level inside the level loop left partial lines behind, the fill-pass
// lesson of msda.hip).  Phase A of each level then reads its points back like the plain kernel does (L2 hits).
template <typename T>
__device__ __forceinline__ void mw_prepare(const MwRaw& rw, const MsdaLevels& lv, const int* qbase, const int* qlin, const bool* qok, int b,
                                           int Nq, int head) {
  constexpr int G = WinGeom<T>::G;
  const int sub = threadIdx.x % G;
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i) {
    const bool own = sub < 8 && qok[i];
    const T* lp = (const T*)rw.logit + (long)qlin[i] * rw.logit_ld + head * 32 + sub;
    float lg[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) lg[l] = own ? Io<T>::ld(lp + l * 8) : -INFINITY;
    float m = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (!qok[i]) m = 0.f;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) { lg[l] = own ? expf(lg[l] - m) : 0.f; s += lg[l]; }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (own) {
      const T* op = (const T*)rw.off + (long)qlin[i] * rw.off_ld + (head * 32 + sub) * 2;
      const float* rp = rw.ref + (long)b * rw.ref_sb + (long)(qlin[i] - b * Nq) * rw.ref_sq;
      float2 xy[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        float ox, oy;
        mw_ld_pair<T>(op + l * 16, ox, oy);
        xy[l] = make_float2(rp[l * rw.ref_sl] + ox / (float)lv.W[l], rp[l * rw.ref_sl + 1] + oy / (float)lv.H[l]);   // msda_prep_fwd_k's arithmetic
      }
      const long o = (long)qbase[i] + sub;
#pragma unroll
      for (int l = 0; l < 4; ++l) *(float2*)(rw.loc_out + 2 * (o + l * 8)) = xy[l];
#pragma unroll
      for (int l = 0; l < 4; ++l) rw.attw_out[o + l * 8] = lg[l] / s;
    }
  }
}

// Workgroup -> (image, head, tile).  XCD x (= blockIdx % 8) works on a contiguous eighth of the (image, head, tile) range: what
// it has in flight samples one head of one image (4.2 MB of value rows at the KITTI shape ~ its 4 MB L2).
struct MwJob { int b, head, seg, ty, tx; bool live; };
__device__ __forceinline__ MwJob mw_job(const MsdaQGrid& qg, int nH, int B) {
  MwJob j;
  const int ntiles = qg.tile_first[qg.nseg];
  const long idx = msda_xcd_block(blockIdx.x, gridDim.x);
  const long total = (long)B * ntiles * nH;
  j.live = idx < total;
  const long i = j.live ? idx : 0;
  const int tile = (int)(i % ntiles);                         // (image, head, tile): an XCD's resident workgroups share one head
  const long bh = i / ntiles;
  j.head = (int)(bh % nH);
  j.b = (int)(bh / nH);
  int s = 0;
#pragma unroll
  for (int k = 1; k < MSDA_MAX_L; ++k) s += (k < qg.nseg && tile >= qg.tile_first[k]) ? 1 : 0;
  j.seg = s;
  const int t = tile - qg.tile_first[s];
  j.ty = t / qg.tiles_x[s];
  j.tx = t - j.ty * qg.tiles_x[s];
  return j;
}

// Phases A + B for one level.  Each lane with sub < 8 owns sampling point `sub` of the MW_QPG queries of its group:
// px / py = pixel coordinates, pw = attention weight (0 for a point that samples nothing or a query outside the map).
// Returns true when the window was staged; box = {xmin, ymin, width, rows}.
template <typename T, bool RAW = false>
__device__ __forceinline__ bool mw_stage(const T* __restrict__ vl, int Wl, int Hl, int nh64, const float* __restrict__ loc,
                                         const float* __restrict__ attw, const int* qbase, const bool* qok, int l, int P,
                                         float* px, float* py, float* pw, uint4* win, int* s_box, int box[4],
                                         const MwRaw* rw = nullptr, const int* qlin = nullptr, int b = 0, int Nq = 0, int head = 0) {
  constexpr int G = WinGeom<T>::G, NG = WinGeom<T>::NG, CAP = WinGeom<T>::CAP;
  const int sub = threadIdx.x % G;
  int xmin = INT_MAX, ymin = INT_MAX, xmax = -1, ymax = -1;
  if (threadIdx.x == 0) { s_box[0] = INT_MAX; s_box[1] = INT_MAX; s_box[2] = -1; s_box[3] = -1; }
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i) {
    float x = -2.f, y = -2.f, w = 0.f;
    if (sub < 8 && qok[i]) {
      const long o = (long)qbase[i] + l * P + sub;
      if constexpr (RAW) {
        const float2 xy = *(const float2*)(rw->loc_out + 2 * o);      // written by mw_prepare (this lane)
        w = rw->attw_out[o];
        x = xy.x * (float)Wl - 0.5f; y = xy.y * (float)Hl - 0.5f;
      } else {
        const float2 xy = *(const float2*)(loc + 2 * o);
        w = attw[o];
        x = xy.x * (float)Wl - 0.5f; y = xy.y * (float)Hl - 0.5f;
      }
    }
    const bool in = y > -1.f && x > -1.f && y < (float)Hl && x < (float)Wl;     // NaN-safe: a NaN location samples nothing
    if (!in) { x = -2.f; y = -2.f; w = 0.f; }
    px[i] = x; py[i] = y; pw[i] = w;
    if (in) {
      const MwTap t = mw_tap(x, y, Wl, Hl);
      xmin = min(xmin, t.xa); xmax = max(xmax, t.xb); ymin = min(ymin, t.ya); ymax = max(ymax, t.yb);
    }
  }
  xmin = mw_wave_min(xmin); ymin = mw_wave_min(ymin); xmax = mw_wave_max(xmax); ymax = mw_wave_max(ymax);
  __syncthreads();                                            // s_box initialised; previous level's window reads done
  if ((threadIdx.x & 63) == 0 && xmax >= 0) {
    atomicMin(&s_box[0], xmin); atomicMin(&s_box[1], ymin); atomicMax(&s_box[2], xmax); atomicMax(&s_box[3], ymax);
  }
  __syncthreads();
  xmin = s_box[0]; ymin = s_box[1]; xmax = s_box[2]; ymax = s_box[3];
  const bool any = xmax >= 0;
  const int bw = any ? xmax - xmin + 1 : 0, bh = any ? ymax - ymin + 1 : 0;
  const int rows = bw * bh;
  box[0] = any ? xmin : 0; box[1] = any ? ymin : 0; box[2] = bw; box[3] = rows;
  const bool staged = any && rows <= CAP;
  if (staged) {
    const T* src = vl + sub * WinGeom<T>::CPL;
    for (int r = threadIdx.x / G; r < rows; r += NG) {
      const int ry = r / bw, rx = r - ry * bw;
      win[r * G + sub] = *(const uint4*)(src + (long)((ymin + ry) * Wl + xmin + rx) * nh64);
    }
  }
  __syncthreads();
  return staged;
}

template <typename T> struct MwAcc;
template <> struct MwAcc<bf16_t> {
  // four chained FMAs per channel straight into the accumulator (no separate sum + add: the kernel is VALU-bound)
  static __device__ __forceinline__ void fma4(float* acc, const uint4& a, const uint4& b, const uint4& c, const uint4& d,
                                              float w00, float w01, float w10, float w11) {
    const uint32_t ra[4] = {a.x, a.y, a.z, a.w}, rb[4] = {b.x, b.y, b.z, b.w}, rc[4] = {c.x, c.y, c.z, c.w}, rd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = acc[2 * i], hi = acc[2 * i + 1];
      lo = fmaf(w00, __uint_as_float(ra[i] << 16), lo); hi = fmaf(w00, __uint_as_float(ra[i] & 0xffff0000u), hi);
      lo = fmaf(w01, __uint_as_float(rb[i] << 16), lo); hi = fmaf(w01, __uint_as_float(rb[i] & 0xffff0000u), hi);
      lo = fmaf(w10, __uint_as_float(rc[i] << 16), lo); hi = fmaf(w10, __uint_as_float(rc[i] & 0xffff0000u), hi);
      lo = fmaf(w11, __uint_as_float(rd[i] << 16), lo); hi = fmaf(w11, __uint_as_float(rd[i] & 0xffff0000u), hi);
      acc[2 * i] = lo; acc[2 * i + 1] = hi;
    }
  }
};
// bf16 forward with the four tap weights rounded to bf16 and travelling as two bf16x2 dwords (w00 | w01, w10 | w11): per dword of
// a value row (two channels) two v_perm pair the corners up per channel and two v_dot2_f32_bf16 accumulate them — 8 instead of 16
// vector instructions per dword, and two shuffles per point instead of four.  The products are exact and the sums fp32; what is
// given up is 2^-9 relative on each tap weight, below the bf16 rounding of the output the path ends in (fp32 path: exact weights).
__device__ __forceinline__ void mw_fma4_packed(float* acc, const uint4& a, const uint4& b, const uint4& c, const uint4& d, uint32_t wab,
                                               uint32_t wcd) {
  const uint32_t ra[4] = {a.x, a.y, a.z, a.w}, rb[4] = {b.x, b.y, b.z, b.w}, rc[4] = {c.x, c.y, c.z, c.w}, rd[4] = {d.x, d.y, d.z, d.w};
  const bf16x2_t vab = __builtin_bit_cast(bf16x2_t, wab), vcd = __builtin_bit_cast(bf16x2_t, wcd);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t ab_lo = __builtin_amdgcn_perm(rb[i], ra[i], 0x05040100u), ab_hi = __builtin_amdgcn_perm(rb[i], ra[i], 0x07060302u);
    const uint32_t cd_lo = __builtin_amdgcn_perm(rd[i], rc[i], 0x05040100u), cd_hi = __builtin_amdgcn_perm(rd[i], rc[i], 0x07060302u);
    float lo = acc[2 * i], hi = acc[2 * i + 1];
    lo = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ab_lo), vab, lo, false);
    hi = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ab_hi), vab, hi, false);
    lo = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, cd_lo), vcd, lo, false);
    hi = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, cd_hi), vcd, hi, false);
    acc[2 * i] = lo; acc[2 * i + 1] = hi;
  }
}
template <> struct MwAcc<float> {
  static __device__ __forceinline__ void fma4(float* acc, const uint4& a, const uint4& b, const uint4& c, const uint4& d,
                                              float w00, float w01, float w10, float w11) {
    const uint32_t ra[4] = {a.x, a.y, a.z, a.w}, rb[4] = {b.x, b.y, b.z, b.w}, rc[4] = {c.x, c.y, c.z, c.w}, rd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[i];
      v = fmaf(w00, __uint_as_float(ra[i]), v); v = fmaf(w01, __uint_as_float(rb[i]), v);
      v = fmaf(w10, __uint_as_float(rc[i]), v); v = fmaf(w11, __uint_as_float(rd[i]), v);
      acc[i] = v;
    }
  }
};

// Shared prologue: the MW_QPG queries of this lane group.  Local query index g + NG * i inside the TQH x TQW tile.
template <typename T>
__device__ __forceinline__ void mw_queries(const MsdaQGrid& qg, const MwJob& job, int Nq, int nH, int LP, int* qbase, int* qrow, bool* qok,
                                           int* qlin = nullptr) {
  constexpr int G = WinGeom<T>::G, NG = WinGeom<T>::NG, TQW = WinGeom<T>::TQW, TQH = WinGeom<T>::TQH;
  const int g = threadIdx.x / G;
  const int Hs = qg.H[job.seg], Ws = qg.W[job.seg];
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i) {
    const int qi = g + NG * i;
    const int y = job.ty * TQH + qi / TQW, x = job.tx * TQW + qi % TQW;
    qok[i] = job.live && y < Hs && x < Ws;
    const long q = qg.start[job.seg] + (long)(qok[i] ? y : 0) * Ws + (qok[i] ? x : 0);
    if (qlin) qlin[i] = (int)((long)job.b * Nq + q);
    qrow[i] = (int)(((long)job.b * Nq + q) * nH + job.head);    // (b, q, head) index: * 64 = output / gradient row
    qbase[i] = qrow[i] * LP;                                     // * 2 = loc offset, * 1 = attw offset (launcher: < 2^31)
  }
}

// PRE = true: the lane that owns a sampling point also does its tap arithmetic once (window row index with the +1 column /
// +1 row flags packed into bits 30 / 31, the four final weights) and the 8 lanes of the group fetch 5 values per point by
// ds_bpermute, instead of every lane redoing the floor / clamp / mask arithmetic from (x, y, weight): ~35 VALU less per point.
template <typename T, bool PRE, bool RAW = false>
__global__ void __launch_bounds__(256) msda_fwd_win_k(const T* __restrict__ value, MsdaLevels lv, MsdaQGrid qg,
                                                      const float* __restrict__ loc, const float* __restrict__ attw,
                                                      T* __restrict__ out, int B, int Nv, int Nq, int nH, int L, int P, MwRaw rw = MwRaw()) {
  constexpr int CPL = WinGeom<T>::CPL, G = WinGeom<T>::G;
  __shared__ uint4 win[MW_BYTES / 16];
  __shared__ int s_box[4];
  const int sub = threadIdx.x % G;
  const MwJob job = mw_job(qg, nH, B);
  const int nh64 = nH * 64;
  int qbase[MW_QPG], qrow[MW_QPG], qlin[MW_QPG];
  bool qok[MW_QPG];
  mw_queries<T>(qg, job, Nq, nH, L * P, qbase, qrow, qok, RAW ? qlin : nullptr);
  if constexpr (RAW) mw_prepare<T>(rw, lv, qbase, qlin, qok, job.b, Nq, job.head);
  float acc[MW_QPG][CPL];
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[i][c] = 0.f;
  const T* vb = value + ((long)job.b * Nv * nH + job.head) * 64;
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.H[l], Wl = lv.W[l];
    const T* vl = vb + (long)lv.start[l] * nH * 64;
    float px[MW_QPG], py[MW_QPG], pw[MW_QPG];
    int box[4];
    const bool staged = mw_stage<T, RAW>(vl, Wl, Hl, nh64, loc, attw, qbase, qok, l, P, px, py, pw, win, s_box, box, &rw, qlin, job.b, Nq,
                                         job.head);
    if (box[3] == 0) continue;                                  // nothing in this tile samples level l (uniform)
    const int x0w = box[0], y0w = box[1], bw = box[2];
    const uint4* gsrc = (const uint4*)(vl + sub * CPL);
    const int s16 = nh64 * (int)sizeof(T) / 16;                 // global row stride in 16-byte units
    if constexpr (PRE) {
      const int rstride = staged ? bw : Wl;
      int pk[MW_QPG];
      float w00[MW_QPG], w01[MW_QPG], w10[MW_QPG], w11[MW_QPG];
#pragma unroll
      for (int i = 0; i < MW_QPG; ++i) {                        // this lane's own point of query i
        const bool in = px[i] > -1.5f;
        const MwTap t = mw_tap(px[i], py[i], Wl, Hl);
        const float bx = 1.f - t.ax, by = 1.f - t.ay;
        const float wxa = (in && t.kxa) ? bx : 0.f, wxb = (in && t.kxb) ? t.ax : 0.f;
        const float wya = t.kya ? by : 0.f, wyb = t.kyb ? t.ay : 0.f;
        const int i00 = staged ? mul24(t.ya - y0w, bw) + (t.xa - x0w) : mul24(t.ya, Wl) + t.xa;
        pk[i] = in ? (i00 | ((t.xb - t.xa) << 30) | ((t.yb - t.ya) << 31)) : 0;
        w00[i] = wya * wxa * pw[i]; w01[i] = wya * wxb * pw[i]; w10[i] = wyb * wxa * pw[i]; w11[i] = wyb * wxb * pw[i];
      }
#pragma unroll
      for (int i = 0; i < MW_QPG; ++i) {
#pragma unroll 2
        for (int p = 0; p < 8; ++p) {
          const int k = __shfl(pk[i], p, G);
          const bool packed = sizeof(T) == 2;              // bf16 storage path: see mw_fma4_packed
          float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
          uint32_t wab = 0, wcd = 0;
          if (packed) {
            wab = (uint32_t)__shfl((int)((uint32_t)f2bf(w00[i]) | ((uint32_t)f2bf(w01[i]) << 16)), p, G);
            wcd = (uint32_t)__shfl((int)((uint32_t)f2bf(w10[i]) | ((uint32_t)f2bf(w11[i]) << 16)), p, G);
          } else {
            a = __shfl(w00[i], p, G); b = __shfl(w01[i], p, G); c = __shfl(w10[i], p, G); d = __shfl(w11[i], p, G);
          }
          const int i00 = k & 0x3fffffff, dx = (k >> 30) & 1;
          const int i10 = i00 + ((k >> 31) & rstride);
          uint4 r00, r01, r10, r11;
          if (staged) {
            r00 = win[i00 * G + sub]; r01 = win[(i00 + dx) * G + sub];
            r10 = win[i10 * G + sub]; r11 = win[(i10 + dx) * G + sub];
          } else {
            r00 = gsrc[mul24(i00, s16)]; r01 = gsrc[mul24(i00 + dx, s16)];
            r10 = gsrc[mul24(i10, s16)]; r11 = gsrc[mul24(i10 + dx, s16)];
          }
          if (packed) mw_fma4_packed(acc[i], r00, r01, r10, r11, wab, wcd);
          else MwAcc<T>::fma4(acc[i], r00, r01, r10, r11, a, b, c, d);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MW_QPG; ++i) {
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
          const float x = __shfl(px[i], p, G), y = __shfl(py[i], p, G), wgt = __shfl(pw[i], p, G);
          const bool in = x > -1.5f;
          const MwTap t = mw_tap(x, y, Wl, Hl);
          const float bx = 1.f - t.ax, by = 1.f - t.ay;
          const float wxa = (in && t.kxa) ? bx : 0.f, wxb = (in && t.kxb) ? t.ax : 0.f;
          const float wya = t.kya ? by : 0.f, wyb = t.kyb ? t.ay : 0.f;
          uint4 r00, r01, r10, r11;
          if (staged) {
            // a point that samples nothing reads row 0 of the window (valid data) with zero weights
            const int ra = in ? (t.ya - y0w) * bw - x0w : 0, rb = in ? (t.yb - y0w) * bw - x0w : 0;
            const int xa = in ? t.xa : 0, xb = in ? t.xb : 0;
            r00 = win[(ra + xa) * G + sub]; r01 = win[(ra + xb) * G + sub];
            r10 = win[(rb + xa) * G + sub]; r11 = win[(rb + xb) * G + sub];
          } else {
            r00 = gsrc[mul24(mul24(t.ya, Wl) + t.xa, s16)]; r01 = gsrc[mul24(mul24(t.ya, Wl) + t.xb, s16)];
            r10 = gsrc[mul24(mul24(t.yb, Wl) + t.xa, s16)]; r11 = gsrc[mul24(mul24(t.yb, Wl) + t.xb, s16)];
          }
          MwAcc<T>::fma4(acc[i], r00, r01, r10, r11, wya * wxa * wgt, wya * wxb * wgt, wyb * wxa * wgt, wyb * wxb * wgt);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i)
    if (qok[i]) VecL<T>::st(out + (long)qrow[i] * 64 + sub * CPL, acc[i]);
}

// d_loc / d_attw: per point the four <gradient row, value row> dot products (raw bf16 pairs through v_dot2c), the three
// sums (weight, d/dx, d/dy) reduce-scattered over the lane group exactly as in msda_bwd_lw_k.
template <typename T, bool PRE>
__global__ void __launch_bounds__(256) msda_bwd_lw_win_k(const T* __restrict__ value, MsdaLevels lv, MsdaQGrid qg,
                                                         const float* __restrict__ loc, const float* __restrict__ attw,
                                                         const T* __restrict__ gout, float* __restrict__ d_loc,
                                                         float* __restrict__ d_attw, int B, int Nv, int Nq, int nH, int L, int P) {
  constexpr int CPL = WinGeom<T>::CPL, G = WinGeom<T>::G;
  __shared__ uint4 win[MW_BYTES / 16];
  __shared__ int s_box[4];
  const int sub = threadIdx.x % G;
  const MwJob job = mw_job(qg, nH, B);
  const int nh64 = nH * 64;
  const int LP = L * P;
  int qbase[MW_QPG], qrow[MW_QPG];
  bool qok[MW_QPG];
  mw_queries<T>(qg, job, Nq, nH, LP, qbase, qrow, qok);
  lw_raw_t go[MW_QPG];
#pragma unroll
  for (int i = 0; i < MW_QPG; ++i) go[i] = *(const lw_raw_t*)(gout + (long)qrow[i] * 64 + sub * CPL);   // clamped to a valid row when !qok
  const T* vb = value + ((long)job.b * Nv * nH + job.head) * 64;
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.H[l], Wl = lv.W[l];
    const T* vl = vb + (long)lv.start[l] * nH * 64;
    float px[MW_QPG], py[MW_QPG], pw[MW_QPG];
    int box[4];
    const bool staged = mw_stage<T>(vl, Wl, Hl, nh64, loc, attw, qbase, qok, l, P, px, py, pw, win, s_box, box);
    const bool empty = box[3] == 0;                             // uniform: every gradient of this level is zero
    const int x0w = box[0], y0w = box[1], bw = box[2];
    const lw_raw_t* gsrc = (const lw_raw_t*)(vl + sub * CPL);
    const lw_raw_t* lwin = (const lw_raw_t*)win;
    const int s16 = nh64 * (int)sizeof(T) / 16;
    const int rstride = staged ? bw : Wl;
    int pk[MW_QPG];
    if constexpr (PRE) {
      // owner lane: window row index (26 bits) | corner masks kxa, kxb, kya, kyb (bits 26-29) | +1 column / +1 row flags (30, 31);
      // px / py are overwritten with the fractional parts, pw stays the weight
#pragma unroll
      for (int i = 0; i < MW_QPG; ++i) {
        const bool in = px[i] > -1.5f;
        const MwTap t = mw_tap(px[i], py[i], Wl, Hl);
        const int i00 = staged ? mul24(t.ya - y0w, bw) + (t.xa - x0w) : mul24(t.ya, Wl) + t.xa;
        const int m = (t.kxa ? 1 : 0) | (t.kxb ? 2 : 0) | (t.kya ? 4 : 0) | (t.kyb ? 8 : 0);
        pk[i] = in ? (i00 | (m << 26) | ((t.xb - t.xa) << 30) | ((t.yb - t.ya) << 31)) : 0;
        px[i] = t.ax; py[i] = t.ay;
      }
    }
#pragma unroll
    for (int i = 0; i < MW_QPG; ++i) {
      float part[24];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float sv = 0.f, sx = 0.f, sy = 0.f;
        if (PRE && !empty) {
          const int k = __shfl(pk[i], p, G);
          const float ax = __shfl(px[i], p, G), ay = __shfl(py[i], p, G), wgt = __shfl(pw[i], p, G);
          const float bx = 1.f - ax, by = 1.f - ay;
          const int i00 = k & 0x03ffffff, dx = (k >> 30) & 1;
          const int i10 = i00 + ((k >> 31) & rstride);
          lw_raw_t r00, r01, r10, r11;
          if (staged) {
            r00 = lwin[i00 * G + sub]; r01 = lwin[(i00 + dx) * G + sub];
            r10 = lwin[i10 * G + sub]; r11 = lwin[(i10 + dx) * G + sub];
          } else {
            r00 = gsrc[mul24(i00, s16)]; r01 = gsrc[mul24(i00 + dx, s16)];
            r10 = gsrc[mul24(i10, s16)]; r11 = gsrc[mul24(i10 + dx, s16)];
          }
          float d00 = RowDot<T>::dot(go[i], r00), d01 = RowDot<T>::dot(go[i], r01);
          float d10 = RowDot<T>::dot(go[i], r10), d11 = RowDot<T>::dot(go[i], r11);
          const bool kxa = k & (1 << 26), kxb = k & (2 << 26), kya = k & (4 << 26), kyb = k & (8 << 26);
          d00 = (kya && kxa) ? d00 : 0.f; d01 = (kya && kxb) ? d01 : 0.f;
          d10 = (kyb && kxa) ? d10 : 0.f; d11 = (kyb && kxb) ? d11 : 0.f;
          sv = by * bx * d00 + by * ax * d01 + ay * bx * d10 + ay * ax * d11;
          sx = (by * (d01 - d00) + ay * (d11 - d10)) * (wgt * (float)Wl);
          sy = (bx * (d10 - d00) + ax * (d11 - d01)) * (wgt * (float)Hl);
        }
        if (!PRE && !empty) {
          const float x = __shfl(px[i], p, G), y = __shfl(py[i], p, G), wgt = __shfl(pw[i], p, G);
          const bool in = x > -1.5f;
          const MwTap t = mw_tap(x, y, Wl, Hl);
          const float bx = 1.f - t.ax, by = 1.f - t.ay;
          lw_raw_t r00, r01, r10, r11;
          if (staged) {
            const int ra = in ? (t.ya - y0w) * bw - x0w : 0, rb = in ? (t.yb - y0w) * bw - x0w : 0;
            const int xa = in ? t.xa : 0, xb = in ? t.xb : 0;
            r00 = lwin[(ra + xa) * G + sub]; r01 = lwin[(ra + xb) * G + sub];
            r10 = lwin[(rb + xa) * G + sub]; r11 = lwin[(rb + xb) * G + sub];
          } else {
            r00 = gsrc[mul24(mul24(t.ya, Wl) + t.xa, s16)]; r01 = gsrc[mul24(mul24(t.ya, Wl) + t.xb, s16)];
            r10 = gsrc[mul24(mul24(t.yb, Wl) + t.xa, s16)]; r11 = gsrc[mul24(mul24(t.yb, Wl) + t.xb, s16)];
          }
          float d00 = RowDot<T>::dot(go[i], r00), d01 = RowDot<T>::dot(go[i], r01);
          float d10 = RowDot<T>::dot(go[i], r10), d11 = RowDot<T>::dot(go[i], r11);
          const bool k_xa = in && t.kxa, k_xb = in && t.kxb;
          d00 = (t.kya && k_xa) ? d00 : 0.f; d01 = (t.kya && k_xb) ? d01 : 0.f;
          d10 = (t.kyb && k_xa) ? d10 : 0.f; d11 = (t.kyb && k_xb) ? d11 : 0.f;
          sv = by * bx * d00 + by * t.ax * d01 + t.ay * bx * d10 + t.ay * t.ax * d11;
          sx = (by * (d01 - d00) + t.ay * (d11 - d10)) * (wgt * (float)Wl);
          sy = (bx * (d10 - d00) + t.ax * (d11 - d01)) * (wgt * (float)Hl);
        }
        part[p] = sv; part[8 + p] = sx; part[16 + p] = sy;
      }
      // 24 partial sums per (query, level), reduce-scattered over the lane group (24 -> 12 -> 6 -> 3 values per lane)
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
      if (qok[i] && (G == 8 || (sub & 1) == 0)) {
        const int base = ((sub / (G / 2)) & 1) * 12 + ((sub / (G / 4)) & 1) * 6 + ((sub / (G / 8)) & 1) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int idx = base + k;
          const int which = idx >> 3, j = l * 8 + (idx & 7);
          if (which == 0) d_attw[(long)qbase[i] + j] = part[k];
          else d_loc[((long)qbase[i] + j) * 2 + (which - 1)] = part[k];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------ host side
static int mw_qgrid(const int* query_hw, int n_qseg, int Nq, int tqh, int tqw, MsdaQGrid& qg) {
  if (!query_hw || n_qseg <= 0 || n_qseg > MSDA_MAX_L) return GE_ERR_UNSUPPORTED;
  qg.nseg = n_qseg;
  long start = 0;
  int n = 0;
  for (int s = 0; s < MSDA_MAX_L; ++s) {
    const bool used = s < n_qseg;
    qg.H[s] = used ? query_hw[2 * s] : 1; qg.W[s] = used ? query_hw[2 * s + 1] : 1; qg.start[s] = (int)start;
    if (qg.H[s] <= 0 || qg.W[s] <= 0) return GE_ERR_BAD_ARG;
    qg.tile_first[s] = n;
    qg.tiles_x[s] = (qg.W[s] + tqw - 1) / tqw;
    if (used) {
      start += (long)qg.H[s] * qg.W[s];
      n += qg.tiles_x[s] * ((qg.H[s] + tqh - 1) / tqh);
    }
  }
  qg.tile_first[MSDA_MAX_L] = n;
  for (int s = n_qseg; s <= MSDA_MAX_L; ++s) qg.tile_first[s] = n;
  return start == Nq ? GE_OK : GE_ERR_BAD_ARG;
}

// The window kernels handle P == 8 (the HAHI configuration) and query geometries given as (H, W) segments; a query set
// without geometry is tiled as one row, which only loses the locality (the global fallback then serves most tiles).
int msda_win_supported(int B, int Nq, int nH, int L, int P, int Nv) {
  return P == 8 && L >= 1 && L <= MSDA_MAX_L && (long)Nv * nH * 64 < (1L << 31) && (long)B * Nq * nH * L * P < (1L << 31);
}

int msda_fwd_win_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const float* loc, const float* attw,
                        void* out, int B, int Nv, int Nq, int nH, int L, int P, int dtype, bool pre, hipStream_t s) {
  MsdaQGrid qg;
  const bool f32 = dtype == GE_F32;
  int e = mw_qgrid(query_hw, n_qseg, Nq, f32 ? WinGeom<float>::TQH : WinGeom<bf16_t>::TQH, 16, qg);
  if (e) return e;
  const long total = (long)B * qg.tile_first[qg.nseg] * nH;
  if (total <= 0) return GE_OK;
  if (total > (1L << 30)) return GE_ERR_UNSUPPORTED;
  const unsigned blocks = msda_grid(total, 1);
#define MW_FWD(TT, PR) msda_fwd_win_k<TT, PR><<<blocks, 256, 0, s>>>((const TT*)value, lv, qg, loc, attw, (TT*)out, B, Nv, Nq, nH, L, P)
  if (f32) { if (pre) MW_FWD(float, true); else MW_FWD(float, false); }
  else { if (pre) MW_FWD(bf16_t, true); else MW_FWD(bf16_t, false); }
#undef MW_FWD
  GE_LAUNCH_CHECK();
  return GE_OK;
}

// Fused prepare + forward (RAW): owner-lane kernel only (PRE), L == 4 and P == 8 (the HAHI configuration; checked by the caller).
int msda_fwd_win_raw_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const MwRaw& rw, void* out, int B,
                            int Nv, int Nq, int nH, int L, int P, int dtype, hipStream_t s) {
  MsdaQGrid qg;
  const bool f32 = dtype == GE_F32;
  int e = mw_qgrid(query_hw, n_qseg, Nq, f32 ? WinGeom<float>::TQH : WinGeom<bf16_t>::TQH, 16, qg);
  if (e) return e;
  const long total = (long)B * qg.tile_first[qg.nseg] * nH;
  if (total <= 0) return GE_OK;
  if (total > (1L << 30)) return GE_ERR_UNSUPPORTED;
  const unsigned blocks = msda_grid(total, 1);
  if (f32) msda_fwd_win_k<float, true, true><<<blocks, 256, 0, s>>>((const float*)value, lv, qg, nullptr, nullptr, (float*)out, B, Nv, Nq, nH, L, P, rw);
  else msda_fwd_win_k<bf16_t, true, true><<<blocks, 256, 0, s>>>((const bf16_t*)value, lv, qg, nullptr, nullptr, (bf16_t*)out, B, Nv, Nq, nH, L, P, rw);
  GE_LAUNCH_CHECK();
  return GE_OK;
}

int msda_bwd_lw_win_launch(const void* value, const MsdaLevels& lv, const int* query_hw, int n_qseg, const float* loc,
                           const float* attw, const void* gout, float* d_loc, float* d_attw, int B, int Nv, int Nq, int nH, int L,
                           int P, int dtype, bool pre, hipStream_t s) {
  MsdaQGrid qg;
  const bool f32 = dtype == GE_F32;
  int e = mw_qgrid(query_hw, n_qseg, Nq, f32 ? WinGeom<float>::TQH : WinGeom<bf16_t>::TQH, 16, qg);
  if (e) return e;
  const long total = (long)B * qg.tile_first[qg.nseg] * nH;
  if (total <= 0) return GE_OK;
  if (total > (1L << 30)) return GE_ERR_UNSUPPORTED;
  const unsigned blocks = msda_grid(total, 1);
#define MW_LW(TT, PR) msda_bwd_lw_win_k<TT, PR><<<blocks, 256, 0, s>>>((const TT*)value, lv, qg, loc, attw, (const TT*)gout, d_loc, d_attw, B, Nv, Nq, nH, L, P)
  if (f32) { if (pre) MW_LW(float, true); else MW_LW(float, false); }
  else { if (pre) MW_LW(bf16_t, true); else MW_LW(bf16_t, false); }
#undef MW_LW
  GE_LAUNCH_CHECK();
  return GE_OK;
}
