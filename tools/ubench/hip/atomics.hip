// microbenchmark: throughput of 256-byte coalesced fp32 atomic bursts at random line addresses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void k(float* buf, unsigned nlines, int iters, float v) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int i = 0; i < iters; ++i) {
    unsigned line = hash(wave * 7919u + i * 104729u) % nlines;     // wave-uniform random 256-byte line
    float* p = buf + (size_t)line * 64 + lane;
    if (MODE == 0) atomicAdd(p, v);
    else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 2) { *p = *p + v; }
    else if (MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else if (MODE == 4) { float x = __builtin_nontemporal_load(p); __builtin_nontemporal_store(x + v, p); }
  }
}

template <int MODE> float run(float* buf, size_t bytes, int iters) {
  unsigned nlines = (unsigned)(bytes / 256);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 8, threads = 256;
  k<MODE><<<blocks, threads>>>(buf, nlines, 4, 1.f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE><<<blocks, threads>>>(buf, nlines, iters, 1.f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double n = (double)blocks * threads / 64 * iters;
  printf("mode %d  buf %6zu MB : %8.3f ms  %7.2f G wave-bursts/s  %7.1f GB/s payload\n", MODE, bytes >> 20, ms, n / ms / 1e6, n * 256 / ms / 1e6);
  return ms;
}

int main() {
  size_t big = 1024ull << 20;
  float* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 0, big));
  for (size_t bytes : {big, (size_t)(512ull << 20), (size_t)(128ull << 20), (size_t)(16ull << 20), (size_t)(2ull << 20)}) {
    run<0>(buf, bytes, 512); run<1>(buf, bytes, 512); run<3>(buf, bytes, 512); run<2>(buf, bytes, 512); run<4>(buf, bytes, 512);
  }
  return 0;
}
