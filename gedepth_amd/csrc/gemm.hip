// Token GEMM on the matrix cores (gfx950, bf16 storage, fp32 accumulation): C[M, N] = epi(A[M, K] B[N, K]^T + bias[N]).
// The Linear layers of the hot path are all of this form with M = tokens (5e4 - 8e5) and N, K in {96 ... 3072}: Swin qkv / proj / FFN
// (depth/models/backbones/depthformer_swin.py:193,221,451-459), the deformable-attention projections of the HAHI neck
// (depth/models/necks/hahi.py:279-289,316-325, mmcv MultiScaleDeformableAttention value_proj / sampling_offsets + attention_weights /
// output_proj) and their input gradients dX = dY W (the same kernel on the transposed weight shadow).  The libraries run these
// skinny shapes at 30 - 42 % of the bf16 MFMA peak and 33 - 45 % of HBM at the same time (profiles/r3_library_roofline.txt).
//
// Decomposition (wave64-first, one persistent 512-thread workgroup per CU):
//   tile          = 256 tokens x 256 output features, K walked in steps of 64; wave (wm, wn) of the 2 x 4 wave grid owns 128 tokens x
//                   64 features = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 registers), computed TRANSPOSED
//                   (D[feature][token] = W-fragment x X-fragment): a lane ends up with runs of 4 consecutive FEATURES of one token
//   staging       = LDS-DMA (global_load_lds, 16 B per lane, 1 KB = 8 rows x 128 B per wave-instruction) into 32 KB slots: THREE for the
//                   A operand (tokens: streamed from HBM, the DMA of step g + 2 is issued during step g), TWO for B (weights: every
//                   workgroup reads the same rows out of L2) = all 160 KB of the CU's LDS.  One barrier per K-step; the wait before it
//                   is vmcnt(4) — everything but the four youngest DMA instructions (A of step g + 2) — so the memory queue never drains
//                   at a step boundary, and the steps of the NEXT tile are already in flight while a tile's epilogue runs
//   LDS image     = [row][8 pieces of 16 B] with piece ^= (row >> 1) & 7 applied on the SOURCE address (the DMA writes lane-linear):
//                   the 16-byte fragment reads of 32 consecutive rows at one K offset then cover all 64 banks evenly (conflict-free for
//                   ds_read_b128's 16-lane groups); both operands are K-contiguous in HBM, so fragments are plain ds_read_b128, read
//                   one K group (16) ahead of the MFMAs that use them
//   epilogue      = bias (through the scalar cache) -> bf16 -> a wave-private LDS slot in the A / B slots the last K-step released ->
//                   16-byte stores of 8 FULL 128-byte lines per instruction (stored straight from the accumulator layout the store
//                   path takes ~7 B per cycle and CU: a third of the tile time).  The LDS traffic of the epilogue and the bias loads are
//                   inline asm: for LDS / vector-memory accesses it can see the compiler inserts vmcnt(0) (it cannot tell the
//                   epilogue slot from the slots the in-flight DMA is writing) and that would drain the A ring once per tile
//   schedule      = XCD x (blockIdx % 8) owns the token tiles mt = x (mod 8) and walks (mt, nt) with nt fastest: the N / 256 workgroups
//                   that share a token tile run side by side on ONE L2, so A crosses HBM once; B (<= 4.7 MB) lives in every L2
//   tails         = rows >= M / features >= N are clamped on the load side and masked on the store side; K needs K % 8 == 0 only: pieces
//                   past K are fetched from a 16-byte zero block
// Where it stands (MI355X, tools/ubench/gemm_time.py, profiles/r4_gemm_time.txt): 1.2 - 1.6x the tuned hipBLASLt / rocBLAS solution on the
// skinny shapes (K or N <= 288: 197120 x 96 -> 288 61 vs 85 us, 288 -> 96 62 vs 101), parity +- 10 % on 261800 x 512 -> 512 / 768,
// 7 - 14 % BEHIND on the 788480-row problems and far behind below ~50 tiles (no split-K): the host side picks per shape.
// Diagnosis of the large shapes (build variants GM_DIAG, tools/ubench/gemm/): MFMA + fragment reads alone 385 us (1.6 PFLOP/s), the
// DMA + stores alone 640 us, together 845 us — the kernel is bound by the CU's global->LDS path (7.6 TB/s chip-wide for 4.8 GB of
// tile traffic), not by the matrix cores, and the two overlap badly because all eight waves run in phase.
// Epilogue: bias add in fp32 before the single rounding to bf16 (what addmm's epilogue does).  A fused exact-erf GELU (second output) was
// built and dropped: its temporaries push the K loop's long-lived values into scratch (76 - 112 spilled registers in every variant tried).
#include "common.h"

typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
typedef float gm_f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned gm_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));

#define GM_BM 256
#define GM_BN 256
#define GM_BK 64
#define GM_TILE (GM_BM * GM_BK * 2)          // bytes of one operand tile (32 KB)
#ifndef GM_DIAG
#define GM_DIAG 0                            // measurement aid: 1 no DMA in the K loop, 2 no MFMA, 4 no epilogue
#endif
#define GM_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GM_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __attribute__((aligned(16))) uint32_t gm_zero16[4] = {0, 0, 0, 0};

struct GemmArgs {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  const float* bias;
  bf16_t* C; long ldc;
  long M; int N, K, ntm, ntn;
};

struct GmTile { int mt, nt; };

// q-th tile of this workgroup's XCD: nt fastest
__device__ __forceinline__ GmTile gm_tile(int q, int xcd, int ntn) {
  GmTile t; const int ml = q / ntn; t.nt = q - ml * ntn; t.mt = ml * 8 + xcd; return t;
}

template <int EPI>
__global__ void __launch_bounds__(512, 1) gemm_nt_k(GemmArgs a) {
  // 160 KB: three A slots (ring) + two B slots, 32 KB each; one workgroup per CU
  __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * GM_TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int l31 = lane & 31, h = lane >> 5;

  // schedule: this workgroup's tiles are q = j, j + per, ... of its XCD's list; its K-steps are numbered g = 0 .. total - 1 across tiles
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int cm = (a.ntm - xcd + 7) >> 3;
  const int nq = cm * a.ntn;
  const int nks = (a.K + GM_BK - 1) / GM_BK;
  if (j >= nq) return;
  const int ntl = (nq - j + per - 1) / per;               // tiles of this workgroup
  const int total = ntl * nks;

  // staging role: chunk c = 8 rows r = wv * 32 + c * 8 + (lane >> 3) of the tile, physical piece lane & 7
  const int srow = wv * 32 + (lane >> 3);
  const int pp = lane & 7;
  // fragment read offsets inside a slot (bytes): token rows wm * 128 + tb * 32 + l31, feature rows wn * 64 + nb * 32 + l31;
  // K group kk reads piece (2 kk + h) ^ swz = (h ^ swz) ^ (2 kk): one XOR on the byte offset per group
  const int swz = (lane >> 1) & 7;
  const int xoff0 = (wm * 128 + l31) * 128 + ((h ^ swz) << 4);
  const int woff0 = (wn * 64 + l31) * 128 + ((h ^ swz) << 4);

  gm_f32x16 acc[4][2];

  // LDS-DMA.  The A operand streams from HBM (each token tile is read once per chip, latency ~2 us under load), the B operand from L2
  // (every workgroup reads the same weight rows): A gets a ring of THREE slots (the DMA of step g + 2 is issued during step g), B two.
  // The wait at the end of a step is vmcnt(4): everything but the four youngest DMA instructions (= A of step g + 2) has landed, so
  // the memory queue never runs dry at a step boundary.  Per step a wave issues 4 B + 4 A instructions, spread over the four MFMA groups.
  unsigned sa[4], sb[4];                                  // byte offsets (from A / B) of the staging cursors' rows (tile of step g + 2 / g + 1):
                                                          // 32-bit offsets on a uniform base keep 8 registers out of the K loop
  int lpk[4];                                             // logical K piece (elements) of this lane in chunk c
#pragma unroll
  for (int c = 0; c < 4; ++c) lpk[c] = (pp ^ ((c * 4 + (lane >> 4)) & 7)) * 8;
  auto point_a = [&](int tl) {                            // tl-th tile of this workgroup (clamped: DMA past the end re-stages the last tile)
    const GmTile t = gm_tile(j + (tl < ntl ? tl : ntl - 1) * per, xcd, a.ntn);
    const long m0 = (long)t.mt * GM_BM;
#pragma unroll
    for (int c = 0; c < 4; ++c) { long gm = m0 + srow + c * 8; if (gm > a.M - 1) gm = a.M - 1; sa[c] = (unsigned)((gm * a.lda + lpk[c]) * 2); }
  };
  auto point_b = [&](int tl) {
    const GmTile t = gm_tile(j + (tl < ntl ? tl : ntl - 1) * per, xcd, a.ntn);
    const int n0 = t.nt * GM_BN;
#pragma unroll
    for (int c = 0; c < 4; ++c) { int gn = n0 + srow + c * 8; if (gn > a.N - 1) gn = a.N - 1; sb[c] = (unsigned)(((long)gn * a.ldb + lpk[c]) * 2); }
  };
  auto dma = [&](const bf16_t* basep, unsigned rowoff, int c, int ks, unsigned char* slot) {
    const int k0 = ks * GM_BK;
    const unsigned char* p = (k0 + lpk[c] >= a.K) ? (const unsigned char*)gm_zero16      // K tail: pieces past K come from the zero block
                                                  : (const unsigned char*)basep + (size_t)(rowoff + (unsigned)(k0 * 2));
    // inline asm, not __builtin_amdgcn_global_load_lds: the compiler books a FLAT-class instruction on BOTH counters, and with one of
    // them among the ds_reads every LDS wait of the K loop becomes lgkmcnt(0) — the fragment reads issued a moment ago included.  The
    // hardware counts an LDS-DMA on vmcnt only; all vmcnt waits of this kernel are explicit.
    const unsigned ldst = (unsigned)(uintptr_t)GM_LDS(unsigned char, slot + wv * 4096 + c * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(ldst), "v"(p) : "memory", "m0");
  };
  unsigned char* const aslots = smem;
  unsigned char* const bslots = smem + 3 * GM_TILE;

  // cursors: (tile, K-step) of the next A / B DMA, slot indices
  int a_tl = 0, a_ks = 0, a_slot = 0, b_tl = 0, b_ks = 0, b_slot = 0;
  auto adv_a = [&]() { a_slot = a_slot == 2 ? 0 : a_slot + 1; if (++a_ks == nks) { a_ks = 0; ++a_tl; point_a(a_tl); } };
  auto adv_b = [&]() { b_slot ^= 1; if (++b_ks == nks) { b_ks = 0; ++b_tl; point_b(b_tl); } };
  point_a(0); point_b(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) dma(a.B, sb[c], c, 0, bslots);                           // B(0)
  adv_b();
#pragma unroll
  for (int c = 0; c < 4; ++c) dma(a.A, sa[c], c, 0, aslots);                           // A(0)
  adv_a();
#pragma unroll
  for (int c = 0; c < 4; ++c) dma(a.A, sa[c], c, a_ks, aslots + GM_TILE);              // A(1)
  adv_a();
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                 // B(0), A(0) landed

  int ca = 0, cb = 0, ks = 0, tl = 0;                                             // slots / K-step / tile of the step being computed
#pragma unroll
  for (int tb = 0; tb < 4; ++tb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[tb][nb] = 0.f;
  for (int g = 0; g < total; ++g) {
    // every wave has waited for its own DMA pieces of this step (end of the previous step) and is done reading the slots refilled now
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* xa = aslots + ca * GM_TILE;
    const unsigned char* wb = bslots + cb * GM_TILE;
    unsigned char* da = aslots + a_slot * GM_TILE;
    unsigned char* db = bslots + b_slot * GM_TILE;
    // fragments of K group kk + 1 are read while the MFMAs of group kk run (two register sets)
    gm_bf16x8 wf[2][2], xf[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wf[0][nb] = *(const gm_bf16x8*)(wb + woff0 + nb * 4096);
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) xf[0][tb] = *(const gm_bf16x8*)(xa + xoff0 + tb * 4096);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cs = kk & 1, ns = cs ^ 1;
      if (kk < 3) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[ns][nb] = *(const gm_bf16x8*)(wb + (woff0 ^ ((kk + 1) << 5)) + nb * 4096);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) xf[ns][tb] = *(const gm_bf16x8*)(xa + (xoff0 ^ ((kk + 1) << 5)) + tb * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);                      // keep the reads of group kk + 1 AHEAD of the MFMAs of group kk
      if (!(GM_DIAG & 1)) {                                   // B first: the four youngest DMA instructions at the end of the step are A's
        if (kk < 2) { dma(a.B, sb[2 * kk], 2 * kk, b_ks, db); dma(a.B, sb[2 * kk + 1], 2 * kk + 1, b_ks, db); }
        else { dma(a.A, sa[2 * kk - 4], 2 * kk - 4, a_ks, da); dma(a.A, sa[2 * kk - 3], 2 * kk - 3, a_ks, da); }
      }
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          if (!(GM_DIAG & 2)) acc[tb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cs][nb], xf[cs][tb], acc[tb][nb], 0, 0, 0);
          else acc[tb][nb][0] += (float)wf[cs][nb][0] * (float)xf[cs][tb][0];
    }
    if (!(GM_DIAG & 1)) { adv_b(); adv_a(); }
    // A(g + 1) (issued during step g - 1) and B(g + 1) must have landed before the next barrier; A(g + 2) stays in flight.  Counted from
    // the young end of the queue, so the epilogue stores of an earlier tile (however many the tail masks let through) do not matter.
    // (A hint stream that touches the A lines of the steps ahead with one global_load_dword per wave was tried to shorten the DMA latency:
    // 843 -> 890 us on 788480 x 512 -> 768.  Memory instructions of a wave retire IN ORDER, so a slow hint load holds back the
    // completion of every younger L2-hit DMA in the same queue.)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    const int fa = ca, fb = cb;                               // the slots this step has released
    ca = ca == 2 ? 0 : ca + 1; cb ^= 1;
    if (++ks < nks) continue;
    ks = 0;

    // epilogue.  D layout of acc[tb][nb]: token = l31 (column), feature = 8 (r >> 2) + 4 h + (r & 3) (row) of the 32 x 32 block, i.e. a lane
    // holds runs of 4 consecutive features of ONE token.  Stored straight from the registers those are 8- / 32-byte pieces of 32 different
    // lines per instruction and the store path takes them at ~7 B per cycle and CU (measured: 7.8 us per 128 KB tile, a third of the tile
    // time); through a wave-private LDS slot ([64 tokens][128 B], 16-byte pieces XOR-swizzled by the token) every store instruction writes
    // 8 FULL 128-byte lines.  The slots live in the A and B slots the last K-step has just released (waves 0-3 / 4-7), hence the
    // barrier: every wave must be done READING them.  The DMA of the next steps runs into the other slots meanwhile.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!(GM_DIAG & 4) || tl + 1 >= ntl) {
      // lane-derived offsets of the epilogue are rebuilt here from an opaque copy of the lane id: hoisted out of the K loop as loop
      // invariants (what the compiler does otherwise) they cost ~30 registers inside it and spill
      int lane_e = lane; asm volatile("" : "+v"(lane_e));
      const int l31 = lane_e & 31, h = lane_e >> 5;
      unsigned char* slot = (wv < 4 ? aslots + fa * GM_TILE : bslots + fb * GM_TILE) + (wv & 3) * 8192;
      const GmTile cur = gm_tile(j + tl * per, xcd, a.ntn);
      const long m0 = (long)cur.mt * GM_BM + wm * 128;
      const int n0 = cur.nt * GM_BN + wn * 64;
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));
      const unsigned slot_lds = (unsigned)(uintptr_t)GM_LDS(unsigned char, slot);
      // bias of the wave's 64 features through the SCALAR cache (s_load: lgkmcnt): an ordinary vector load here makes the compiler wait
      // vmcnt(0), which would drain the A ring's in-flight DMA once per tile.  n is wave-uniform; lanes pick their half by h.
      // The bias is added into the accumulators in place (16 bias registers live at a time).
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float bv[16];
        const int n = n0 + nb * 32;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {                                 // 16 features per asm block (16 SGPRs live)
          gm_f32x8 t0 = 0.f, t1 = 0.f;
          if (EPI >= 1 && a.bias && n + 16 * gp + 16 <= a.N) {
            const float* bp = a.bias + n + 16 * gp;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1) : "s"(bp) : "memory");
          } else if (EPI >= 1 && a.bias && n + 16 * gp < a.N) {           // N % 8 == 0: exactly 8 features left
            const float* bp = a.bias + n + 16 * gp;
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : "s"(bp) : "memory");
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) { bv[8 * gp + i] = h ? t0[4 + i] : t0[i]; bv[8 * gp + 4 + i] = h ? t1[4 + i] : t1[i]; }
        }
        if (EPI >= 1) {
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tb][nb][r] += bv[r];
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int tb = half * 2 + t2;
            const int tok = t2 * 32 + l31;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              float v[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = acc[tb][nb][4 * g4 + i];
              gm_u32x2 d;
              d.x = __builtin_bit_cast(unsigned, (b2){(__bf16)v[0], (__bf16)v[1]});
              d.y = __builtin_bit_cast(unsigned, (b2){(__bf16)v[2], (__bf16)v[3]});
              // feature f = nb * 32 + 8 g4 + 4 h: piece f >> 3 = nb * 4 + g4, byte (f & 7) * 2 = 8 h inside it.  Written with an asm
              // ds_write: for a store the compiler can see it waits vmcnt(0) first (it cannot tell this slot from the ones the DMA in
              // flight is writing), and that would drain the A ring once per tile
              const unsigned la = slot_lds + tok * 128 + (((nb * 4 + g4) ^ (tok & 7)) << 4) + 8 * h;
              asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(d) : "memory");
            }
          }
        }
        bf16_t* dstC = a.C;
        const long ldc = a.ldc;
        const long row0 = m0 + half * 64 + (lane_e >> 3);
        const int n = n0 + (lane_e & 7) * 8;
        bf16_t* dp = dstC + row0 * ldc + n;
        const bool inside = m0 + half * 64 + 64 <= a.M && n0 + 64 <= a.N;   // wave-uniform: no per-store masks inside the matrix
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {                          // 4 + 4 rows of 8 tokens: 16 staging registers
          gm_u32x4 o[4];                                          // asm reads for the same reason as the asm writes above
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int tok = (q4 * 4 + it) * 8 + (lane_e >> 3), piece = lane_e & 7;
            const unsigned la = slot_lds + tok * 128 + ((piece ^ (tok & 7)) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(o[it]) : "v"(la) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (inside) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *(gm_u32x4*)(dp + (q4 * 4 + it) * 8 * ldc) = o[it];
          } else {
#pragma unroll
            for (int it = 0; it < 4; ++it)
              if (row0 + (q4 * 4 + it) * 8 < a.M && n < a.N) *(gm_u32x4*)(dp + (q4 * 4 + it) * 8 * ldc) = o[it];
          }
        }
      }
    }
    ++tl;
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[tb][nb] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no DMA may land in this LDS after the workgroup has left
}

extern "C" int ge_gemm_nt(const void* A, long lda, const void* B, long ldb, const float* bias, void* C, long ldc, long M, int N, int K,
                          int dtype, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return GE_ERR_BAD_ARG;
  if (dtype != GE_BF16 || (N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || (ldc & 7)) return GE_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 15)) return GE_ERR_UNSUPPORTED;
  if ((M * lda + K) * 2 >= (1L << 32) || ((long)N * ldb + K) * 2 >= (1L << 32)) return GE_ERR_UNSUPPORTED;      // 32-bit row offsets
  if (M == 0) return GE_OK;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.bias = bias; a.C = (bf16_t*)C; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K;
  const long ntm = (M + GM_BM - 1) / GM_BM;
  if (ntm > (1 << 24)) return GE_ERR_UNSUPPORTED;
  a.ntm = (int)ntm; a.ntn = (N + GM_BN - 1) / GM_BN;
  const int cus = ge_cu_count();                          // per device (common.h)
  if (!cus) return GE_ERR_BAD_ARG;
  int grid = cus & ~7; if (grid < 8) grid = 8;              // one persistent workgroup per CU; workgroups without a tile return at once
  const hipStream_t s = ge_stream(stream);
  if (bias) gemm_nt_k<1><<<grid, 512, 0, s>>>(a);
  else gemm_nt_k<0><<<grid, 512, 0, s>>>(a);
  GE_LAUNCH_CHECK();
  return GE_OK;
}
