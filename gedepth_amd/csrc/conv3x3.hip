// 3x3 / stride 1 / pad 1 convolution on channels-last bf16 maps as an implicit GEMM on the matrix cores (gfx950), with the bias +
// (Leaky)ReLU epilogue of mmcv's ConvModule folded in.  Reference call sites: the UpSample blocks of the DenseDepth head
// (depth/models/decode_heads/densedepth_head.py:14-27), the HAHI fusion convolutions (depth/models/necks/hahi.py:140-165) and the PE-neck
// trunk (necks/pemask_neck.py:36-42); the same kernel with flipped / transposed weights is the data gradient of those layers.
//
//   y[n, y, x, co] = act(b[co] + sum_{r, s, ci} w[co, r, s, ci] x[n, y + r - 1, x + s - 1, ci])        GEMM: M = N H W pixels, N = Cout, K = 9 Cin
//
// Decomposition (wave64-first): a workgroup of 4 waves owns an 8 x 32 pixel tile and 64 output channels; wave w owns rows 2w, 2w + 1 of the
// tile = two 32-pixel M-blocks x two 32-channel N-blocks = four 32x32 accumulators.  K runs over Cin in chunks of 32 channels; per chunk
// the (8 + 2) x (32 + 2) input halo tile and the 9 x 64 weight rows of the chunk are staged in LDS ONCE and all nine taps read them (the
// taps are address offsets into the halo tile, so the input crosses HBM / L2 once per chunk instead of nine times); the next chunk is
// prefetched into registers while the 72 MFMAs of the current one run.  LDS rows are 32 channels + 8 pad (80 bytes): the 16-byte
// A / B fragment reads of 16 neighbouring pixels / channels then fall on distinct banks.  Output: accumulators -> bias / activation ->
// bf16 through LDS -> 16-byte coalesced stores.
//
// v_mfma_f32_32x32x16_bf16 operand maps: A lane (m = lane & 31, kg = lane >> 5) holds A[m][8 kg .. 8 kg + 7] — 8 consecutive input
// channels of pixel m: one ds_read_b128; B lane (n, kg) holds W[n][8 kg ..] — 8 consecutive input channels of output channel n for
// the tap: one ds_read_b128 from the (O, H, W, I) weight layout, which is how a channels-last conv weight is stored anyway.
#include "common.h"
#include <stdlib.h>

typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cv_f32x16 __attribute__((ext_vector_type(16)));

#define CV_TH 8
#define CV_TW 32
#define CV_KC 32                     // input channels per chunk
#define CV_PITCH 40                  // bf16 elements per LDS row (32 + 8 pad)
#define CV_HALO ((CV_TH + 2) * (CV_TW + 2))
#define CV_NT 64                     // output channels per workgroup
#define CV_IN_PIECES (CV_HALO * 4)   // 16-byte pieces of the input halo tile per chunk
#define CV_W_PIECES (9 * CV_NT * 4)

__global__ void __launch_bounds__(256, 2) conv3x3_nhwc_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ y, int H, int W, int Cin, int Cout, int n_ntiles, float slope,
                                                         int act) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[CV_HALO * CV_PITCH + 9 * CV_NT * CV_PITCH];     // 27.2 KB + 46.1 KB: two workgroups per CU
  bf16_t* in_tile = smem;
  bf16_t* w_tile = smem + CV_HALO * CV_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx0 = blockIdx.x * CV_TW, ty0 = blockIdx.y * CV_TH;
  const int n = blockIdx.z / n_ntiles, n0 = (blockIdx.z - n * n_ntiles) * CV_NT;
  const bf16_t* xn = x + (long)n * H * W * Cin;

  // which 16-byte pieces this thread stages (the same for every chunk): source offsets (elements) or -1
  int in_src[6], in_dst[6], w_src[9], w_dst[9];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = tid + i * 256;
    in_src[i] = -1; in_dst[i] = 0;
    if (p < CV_IN_PIECES) {
      const int pix = p >> 2, part = p & 3;
      const int iy = pix / (CV_TW + 2), ix = pix - iy * (CV_TW + 2);
      const int gy = ty0 + iy - 1, gx = tx0 + ix - 1;
      in_dst[i] = pix * CV_PITCH + part * 8;
      in_src[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * Cin + part * 8 : -2;        // -2: inside the tile, outside the image -> zeros
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int p = tid + i * 256;                               // 2304 pieces = 9 * 256: no guard
    const int row = p >> 2, part = p & 3;
    const int tap = row / CV_NT, co = row - tap * CV_NT;
    w_dst[i] = row * CV_PITCH + part * 8;
    w_src[i] = (n0 + co < Cout) ? ((n0 + co) * 9 + tap) * Cin + part * 8 : -2;
  }
  uint4 pin[6], pw[9];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define CV_PREFETCH(C0)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) { pin[i] = zero4; if (in_src[i] >= 0) pin[i] = *(const uint4*)(xn + in_src[i] + (C0)); } \
  _Pragma("unroll") for (int i = 0; i < 9; ++i) { pw[i] = zero4; if (w_src[i] >= 0) pw[i] = *(const uint4*)(w + w_src[i] + (C0)); }
#define CV_PARK()                                                                                \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) if (in_src[i] != -1) *(uint4*)(in_tile + in_dst[i]) = pin[i];             \
  _Pragma("unroll") for (int i = 0; i < 9; ++i) *(uint4*)(w_tile + w_dst[i]) = pw[i];

  cv_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0.f;
  const int m = lane & 31, kg = lane >> 5;
  // A fragment base of M-block mb (tile row 2 wv + mb), tap (0, 0): halo pixel (row, m); B fragment base of N-block nb
  const int a_base0 = ((2 * wv) * (CV_TW + 2) + m) * CV_PITCH + kg * 8;
  const int b_base = m * CV_PITCH + kg * 8;

  CV_PREFETCH(0)
  CV_PARK()
  __syncthreads();
  for (int c0 = 0; c0 < Cin; c0 += CV_KC) {
    const bool more = c0 + CV_KC < Cin;
    if (more) { CV_PREFETCH(c0 + CV_KC) }
    // 18 steps = 9 taps x 2 K groups of 16 channels; the four fragments of step it + 1 are read into the second register set BEFORE the four
    // MFMAs of step it issue (round 4: left to the compiler, every group of MFMAs waited on fragment reads issued a moment earlier —
    // s_waitcnt lgkmcnt(1) / (0) in the middle of each tap)
    {
      cv_bf16x8 A[2][2], B[2][2];
      auto frags = [&](int set, int it) {
        const int tap = it >> 1, ks = it & 1, r = tap / 3, s = tap - 3 * r;
        const int ao = a_base0 + (r * (CV_TW + 2) + s) * CV_PITCH + ks * 16;
        A[set][0] = *(const cv_bf16x8*)(in_tile + ao);
        A[set][1] = *(const cv_bf16x8*)(in_tile + ao + (CV_TW + 2) * CV_PITCH);
        const int bo = b_base + tap * CV_NT * CV_PITCH + ks * 16;
        B[set][0] = *(const cv_bf16x8*)(w_tile + bo);
        B[set][1] = *(const cv_bf16x8*)(w_tile + bo + 32 * CV_PITCH);
      };
      frags(0, 0);
#pragma unroll
      for (int it = 0; it < 18; ++it) {
        if (it < 17) frags((it + 1) & 1, it + 1);
        __builtin_amdgcn_sched_barrier(0);
        const int c = it & 1;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], B[c][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], B[c][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], B[c][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], B[c][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                           // every wave is done reading this chunk
    if (more) { CV_PARK() }
    __syncthreads();
  }
#undef CV_PREFETCH
#undef CV_PARK

  // epilogue: C/D layout: column = lane & 31 (output channel of the N-block), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the M-block).
  // Stage the wave's 64 pixels x 64 channels as bf16 in LDS ([pixel][64 + 8 pad]), then 16-byte stores: 128 contiguous bytes per pixel.
  bf16_t* ot = smem + wv * (64 * 72);                           // 4 waves x 64 x 72 x 2 B = 36.9 KB of the 73 KB; all tile reads are done
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int co = nb * 32 + m;
      const float bv = (bias && n0 + co < Cout) ? bias[n0 + co] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mb][nb][r] + bv;
        if (act) v = v > 0.f ? v : v * slope;
        const int px = (r & 3) + 8 * (r >> 2) + 4 * kg;
        ot[(mb * 32 + px) * 72 + co] = __builtin_bit_cast(bf16_t, (__bf16)v);        // v_cvt_pk_bf16_f32 (round-to-nearest-even, like f2bf)
      }
    }
  __builtin_amdgcn_wave_barrier();
  // the wave's pixels: rows 2 wv, 2 wv + 1 of the tile, 32 columns; lane -> (pixel = it * 8 + lane / 8, 16-byte piece lane % 8)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int p = it * 8 + (lane >> 3), piece = lane & 7;
    const int gy = ty0 + 2 * wv + (p >> 5), gx = tx0 + (p & 31);
    if (gy < H && gx < W && n0 + piece * 8 < Cout)
      *(uint4*)(y + (((long)n * H + gy) * W + gx) * Cout + n0 + piece * 8) = *(const uint4*)(ot + p * 72 + piece * 8);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Second decomposition, for the large maps (round 4): 512-thread workgroup = 16 x 32 pixel tile x 64 output channels, operands staged by
// LDS-DMA into TWO buffers.  What the counters said about the kernel above (profiles/r3_pmc_conv3x3.txt: MFMA busy 35 %, waves waiting on an
// instruction 56 %, bank conflicts 24 % of the LDS cycles) and what the token GEMM taught (csrc/gemm.hip: a CU moves ~12 - 22 B per cycle
// from L2 into LDS, whatever the MFMA rate): per 32-channel chunk a 4-wave workgroup stages 21.7 KB of halo pixels and 36.9 KB of weights
// for 288 MFMAs — 25 B per MFMA-cycle with two workgroups per CU, and 63 % of it is the weight slab every workgroup re-stages.  Here:
//   tile       = 16 rows x 32 pixels: the SAME weight slab serves twice the pixels (halo 18 x 34 pixels = 39.2 KB + weights 36.9 KB per
//                chunk for 576 MFMAs: 16.5 B per MFMA-cycle); wave w owns rows 2 w, 2 w + 1 as before (4 accumulators)
//   staging    = global_load_lds, 16 B per lane, straight into the other buffer while the 72 MFMAs per wave of this chunk run: no register
//                parking, no LDS stores, ONE barrier per chunk; 75 DMA instructions per chunk, 9 - 10 per wave, issued between MFMA groups
//   LDS image  = 64-byte rows (a pixel's / an output channel's 32 input channels) WITHOUT padding — the DMA writes lane-linear — and the
//                16-byte piece index XORed with bits 2-3 of the row index instead (applied on the source address): the fragment reads of
//                16 neighbouring rows then cover all 64 banks.  Out-of-image halo pixels and output channels >= Cout come from a zero block.
// 160 KB of LDS, one workgroup (8 waves) per CU.  Same arithmetic (same products, same fp32 sums in the same order per accumulator).
#define CD_TH 16
#define CD_TW 32
#define CD_HP ((CD_TH + 2) * (CD_TW + 2))                 // 612 halo pixels
#define CD_HCH 39                                         // 1 KB DMA chunks of the halo image (612 * 64 B = 38.25 KB)
#define CD_WCH 36                                         // of the weight image (9 * 64 rows * 64 B)
#define CD_HB (CD_HCH * 1024)
#define CD_BUF (80 * 1024)                                // per buffer: 75 KB of operands + 5 KB that absorb the idle DMA slots (8 waves x 10 slots)
#define CD_NCH (CD_HCH + CD_WCH)                          // 75 DMA instructions per chunk
#define CD_SLOTS 10                                       // per wave: chunks w, w + 8, ...
#define CD_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define CD_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __attribute__((aligned(16))) uint32_t cd_zero16[4] = {0, 0, 0, 0};

__global__ void __launch_bounds__(512, 1) conv3x3_nhwc_dma_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ y, int H, int W, int Cin, int Cout, int n_ntiles, float slope,
                                                             int act) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CD_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx0 = blockIdx.x * CD_TW, ty0 = blockIdx.y * CD_TH;
  const int n = blockIdx.z / n_ntiles, n0 = (blockIdx.z - n * n_ntiles) * CV_NT;
  const unsigned char* xn = (const unsigned char*)(x + (long)n * H * W * Cin);
  const unsigned char* wb = (const unsigned char*)w;

  // DMA sources of this lane: slot i = chunk wv + 8 i; byte offset from xn (halo chunks) / w (weight chunks), ~0u = zero block
  unsigned goff[CD_SLOTS];
#pragma unroll
  for (int i = 0; i < CD_SLOTS; ++i) {
    const int ch = wv + 8 * i;
    unsigned off = ~0u;
    if (ch < CD_HCH) {
      const int q = ch * 64 + lane, hp = q >> 2, piece = (q & 3) ^ ((hp >> 2) & 3);
      if (hp < CD_HP) {
        const int iy = hp / (CD_TW + 2), ix = hp - iy * (CD_TW + 2);
        const int gy = ty0 + iy - 1, gx = tx0 + ix - 1;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) off = (unsigned)(((gy * W + gx) * Cin + piece * 8) * 2);
      }
    } else if (ch < CD_NCH) {
      const int q = (ch - CD_HCH) * 64 + lane, row = q >> 2, piece = (q & 3) ^ ((row >> 2) & 3);
      const int tap = row >> 6, co = row & 63;
      if (n0 + co < Cout) off = (unsigned)((((n0 + co) * 9 + tap) * Cin + piece * 8) * 2);
    }
    goff[i] = off;
  }
  auto dma = [&](int i, int c0, int buf) {                   // slot i of chunk c0 into buffer buf; chunks 75 .. 79 (zero block -> the 5 spare
    const int ch = wv + 8 * i;                               // KB of the buffer) keep the K loop free of branches: with a branch per DMA the
    const unsigned char* base = ch < CD_HCH ? xn : wb;       // compiler waits for ALL outstanding LDS reads at every join
    const unsigned char* p = goff[i] == ~0u ? (const unsigned char*)cd_zero16 : base + (size_t)(goff[i] + (unsigned)(c0 * 2));
    // inline asm instead of __builtin_amdgcn_global_load_lds: see csrc/gemm.hip (the builtin turns every LDS wait near it into lgkmcnt(0))
    const unsigned ldst = (unsigned)(uintptr_t)CD_LDS(unsigned char, smem + buf * CD_BUF + ch * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(ldst), "v"(p) : "memory", "m0");
  };

  cv_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0.f;
  const int m = lane & 31, kg = lane >> 5;
  // fragment addresses (bytes inside a buffer) for ks = 0; ks = 1 is the same address ^ 32 (piece index ^ 2)
  int abase[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int hp = (2 * wv + mb + tap / 3) * (CD_TW + 2) + m + tap % 3;
      abase[tap][mb] = hp * 64 + ((kg ^ ((hp >> 2) & 3)) << 4);
    }
  const int bbase = CD_HB + m * 64 + ((kg ^ ((m >> 2) & 3)) << 4);

#pragma unroll
  for (int i = 0; i < CD_SLOTS; ++i) dma(i, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int buf = 0;
  for (int c0 = 0; c0 < Cin; c0 += CV_KC) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int cn = c0 + CV_KC < Cin ? c0 + CV_KC : c0;        // the last chunk re-stages itself into the idle buffer (unconditional DMA)
    const unsigned char* base = smem + buf * CD_BUF;
    // 18 groups (tap, ks) of 4 reads + 4 MFMAs; the reads of group g + 1 are issued before the MFMAs of group g (two register sets);
    // one DMA instruction of the next chunk after every second group
    cv_bf16x8 A[2][2], B[2][2];
    A[0][0] = *(const cv_bf16x8*)(base + abase[0][0]); A[0][1] = *(const cv_bf16x8*)(base + abase[0][1]);
    B[0][0] = *(const cv_bf16x8*)(base + bbase); B[0][1] = *(const cv_bf16x8*)(base + bbase + 2048);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      const int cs = g & 1, ns = cs ^ 1;
      if (g < 17) {
        const int tap = (g + 1) >> 1, kx = ((g + 1) & 1) << 5;
        A[ns][0] = *(const cv_bf16x8*)(base + (abase[tap][0] ^ kx)); A[ns][1] = *(const cv_bf16x8*)(base + (abase[tap][1] ^ kx));
        B[ns][0] = *(const cv_bf16x8*)(base + (bbase ^ kx) + tap * 4096); B[ns][1] = *(const cv_bf16x8*)(base + (bbase ^ kx) + tap * 4096 + 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(g & 1)) dma(g >> 1, cn, buf ^ 1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cs][0], B[cs][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cs][0], B[cs][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cs][1], B[cs][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cs][1], B[cs][1], acc[1][1], 0, 0, 0);
    }
    dma(9, cn, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    buf ^= 1;
  }
  __builtin_amdgcn_s_barrier();                                  // every wave is done reading the last chunk: the tiles become the output stage
  asm volatile("" ::: "memory");

  // epilogue as in conv3x3_nhwc_k: the wave's 64 pixels x 64 channels as bf16 in LDS ([pixel][64 + 8 pad]), then 16-byte stores
  bf16_t* ot = (bf16_t*)smem + wv * (64 * 72);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int co = nb * 32 + m;
      const float bv = (bias && n0 + co < Cout) ? bias[n0 + co] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mb][nb][r] + bv;
        if (act) v = v > 0.f ? v : v * slope;
        const int px = (r & 3) + 8 * (r >> 2) + 4 * kg;
        ot[(mb * 32 + px) * 72 + co] = __builtin_bit_cast(bf16_t, (__bf16)v);
      }
    }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int p = it * 8 + (lane >> 3), piece = lane & 7;
    const int gy = ty0 + 2 * wv + (p >> 5), gx = tx0 + (p & 31);
    if (gy < H && gx < W && n0 + piece * 8 < Cout)
      *(uint4*)(y + (((long)n * H + gy) * W + gx) * Cout + n0 + piece * 8) = *(const uint4*)(ot + p * 72 + piece * 8);
  }
}

// x (N, H, W, Cin) bf16, w (Cout, 3, 3, Cin) bf16 [the storage order of a channels-last conv weight], bias (Cout) f32 or NULL,
// y (N, H, W, Cout) bf16 = act(conv + bias): act = 0 none, 1 leaky-ReLU with `slope` (0 = ReLU).  Cin % 32 == 0, Cout % 8 == 0.
extern "C" int ge_conv3x3_nhwc_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int Cout, int act,
                                   float slope, int dtype, void* stream) {
  if (!x || !w || !y || N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_BF16 || Cin % CV_KC || Cout % 8 || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15)) return GE_ERR_UNSUPPORTED;
  if ((long)H * W * Cin >= (1L << 31) || (long)Cout * 9 * Cin >= (1L << 31)) return GE_ERR_UNSUPPORTED;
  if (N == 0) return GE_OK;
  const int ntn = (Cout + CV_NT - 1) / CV_NT;
  const char* ev = getenv("GE_CONV3X3");                      // 0 / unset: auto, 1: register-staged kernel, 2: LDS-DMA kernel (A/B timing, tests)
  const int variant = ev ? atoi(ev) : 0;
  // Measured (tools/ubench/conv_time.py, GE_CONV3X3=1 vs 2, profiles/r4_conv_time.txt): the DMA kernel wins where the K loop is long
  // (Cin >= 512: 576 -> 64 @176x560 659 -> 584 us, 1280 -> 768 @11x35 144 -> 113, 608 -> 96 @88x280 273 -> 264), ties at Cin 160 - 300 and
  // loses with two or three chunks per tile (64 -> 64: 84 -> 100 us: one workgroup per CU has nothing to overlap its prologue and
  // epilogue with) or when the 16-row tile adds padding rows (22 x 70).
  const int hpad16 = (H + CD_TH - 1) / CD_TH * CD_TH, hpad8 = (H + CV_TH - 1) / CV_TH * CV_TH;
  const bool dma = variant == 2 || (variant == 0 && Cin >= 512 && hpad16 * 8 <= hpad8 * 9);
  if (dma) {
    const dim3 grid((W + CD_TW - 1) / CD_TW, (H + CD_TH - 1) / CD_TH, N * ntn);
    if (grid.y > 65535 || grid.z > 65535) return GE_ERR_UNSUPPORTED;
    conv3x3_nhwc_dma_k<<<grid, 512, 0, ge_stream(stream)>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, H, W, Cin, Cout, ntn, slope, act);
    GE_LAUNCH_CHECK();
    return GE_OK;
  }
  const dim3 grid((W + CV_TW - 1) / CV_TW, (H + CV_TH - 1) / CV_TH, N * ntn);
  if (grid.y > 65535 || grid.z > 65535) return GE_ERR_UNSUPPORTED;
  conv3x3_nhwc_k<<<grid, 256, 0, ge_stream(stream)>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, H, W, Cin, Cout, ntn, slope, act);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
