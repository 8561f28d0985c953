#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  extern __shared__ float tile[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 256 * 64; i += 256) tile[i] = 0.f;
  __syncthreads();
  unsigned h = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u;
  for (int i = 0; i < iters; ++i) {
    h = hash(h + i);
    const int r = h & 255;                       // wave-uniform row
    if (MODE == 0) atomicAdd(&tile[r * 64 + lane], 1.0f);
    else if (MODE == 1) tile[r * 64 + lane] += 1.0f;
    else if (MODE == 2) __hip_atomic_fetch_add(&tile[r * 64 + lane], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + lane] = tile[lane];
}
template <int MODE> void run(float* out) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 512, iters = 20000;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  k<MODE><<<blocks, 256, 65536>>>(out, 10); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); k<MODE><<<blocks, 256, 65536>>>(out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double n = (double)blocks * 4 * iters;
  printf("mode %d: %.3f ms  %.2f G wave-ops/s  (%.1f cycles/op/CU at 2.1GHz, 2 WG/CU)\n", MODE, ms, n / ms / 1e6, ms * 1e-3 * 2.1e9 * 256 / n);
}
int main() { float* out; CK(hipMalloc(&out, 512 * 64 * 4)); run<0>(out); run<1>(out); run<2>(out); return 0; }
