// Round 5: fp32 row atomics (256 B = 64 channels of one d_value row) when every row is only ever touched from ONE XCD.
// Round 1 (atomics.hip) measured 5.0 G bursts/s chip-wide with every wave hitting random rows of one shared buffer: each line then
// migrates between the eight L2s.  The deformable-attention kernels map XCD x -> image x, so all updates of a d_value row come from one
// XCD; this probe measures what the atomic units deliver in that arrangement, for footprints inside / outside the 4 MB L2.
//   mode 0: shared random rows (round-1 pattern)        mode 1: XCD-private random rows (partition = blockIdx % 8, checked against XCC_ID)
//   mode 2: XCD-private, rows of a moving window (40 consecutive rows, window start random): the flush pattern of a query tile
//   mode 3: as 1 but two 128-byte half rows per instruction (the MFMA C/D layout: lanes 0-31 row r, lanes 32-63 row r + 4)
//   mode 4: as 1 with plain load + store (no atomic): what the memory path alone costs
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics_xcd.hip -o atomics_xcd && ./atomics_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(float* buf, unsigned rows_per_part, int iters, float v, unsigned* xcc_mismatch) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned part = blockIdx.x % 8;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  if (lane == 0 && (xcc & 15) != part) atomicAdd(xcc_mismatch, 1u);
  float* base = buf + (MODE == 0 ? 0 : (size_t)part * rows_per_part * 64);
  const unsigned nrows = MODE == 0 ? rows_per_part * 8 : rows_per_part;
  for (int i = 0; i < iters; ++i) {
    unsigned row;
    if (MODE == 2) row = (hash(wave * 7919u + (i / 40) * 104729u) % (nrows - 40)) + (i % 40);
    else row = hash(wave * 7919u + i * 104729u) % nrows;
    float* p;
    if (MODE == 3) { unsigned r2 = (row & ~7u) | ((row & 3u)) | ((lane >> 5) << 2); p = base + (size_t)r2 * 64 + ((row >> 2) & 1) * 32 + (lane & 31); }
    else p = base + (size_t)row * 64 + lane;
    if (MODE == 4) *p = *p + v;
    else atomicAdd(p, v);
  }
}

template <int MODE> void run(float* buf, size_t bytes_per_part, int iters, unsigned* mm) {
  const unsigned rows = (unsigned)(bytes_per_part / 256);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 8, threads = 256;
  k<MODE><<<blocks, threads>>>(buf, rows, 4, 1.f, mm);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE><<<blocks, threads>>>(buf, rows, iters, 1.f, mm);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double n = (double)blocks * threads / 64 * iters;
  printf("mode %d  %7.2f MB per XCD : %8.3f ms  %7.2f G row-atomics/s  %8.1f GB/s payload\n", MODE, bytes_per_part / 1048576.0, ms, n / ms / 1e6,
         n * 256 / ms / 1e6);
}

int main() {
  const size_t big = 1024ull << 20;
  float* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 0, big));
  unsigned* mm; CK(hipMalloc(&mm, 4)); CK(hipMemset(mm, 0, 4));
  for (size_t per : {(size_t)(1ull << 20), (size_t)(2ull << 20), (size_t)(8ull << 20), (size_t)(64ull << 20), (size_t)(128ull << 20)}) {
    run<0>(buf, per, 512, mm); run<1>(buf, per, 512, mm); run<2>(buf, per, 520, mm); run<3>(buf, per, 512, mm); run<4>(buf, per, 512, mm);
  }
  unsigned h; CK(hipMemcpy(&h, mm, 4, hipMemcpyDeviceToHost));
  printf("workgroups whose XCC_ID != blockIdx %% 8: %u waves (0 = the round-robin mapping holds)\n", h);
  // correctness of XCD-private L2 atomics across a kernel boundary: every row must hold the number of adds it received
  CK(hipMemset(buf, 0, big));
  k<1><<<2048, 256>>>(buf, 4096, 1024, 1.f, mm);
  CK(hipDeviceSynchronize());
  float* hb = (float*)malloc(8ull * 4096 * 256);
  CK(hipMemcpy(hb, buf, 8ull * 4096 * 256, hipMemcpyDeviceToHost));
  double tot = 0; for (size_t i = 0; i < 8ull * 4096 * 64; ++i) tot += hb[i];
  printf("sum of all adds %.0f (expected %.0f)\n", tot, 2048.0 * 4 * 1024 * 64);
  return 0;
}
