// Cost of a 16-byte-per-lane gather instruction (global_load_dwordx4) per CU for different address patterns, from a buffer that fits the
// L2 / Infinity Cache: is the ~29 cycles per instruction of the deformable-attention row gathers a property of the instruction (1 KB per
// wave) or of the number of distinct 128-byte lines it touches?
//   pattern 0: one contiguous 1 KB per instruction (fully coalesced)
//   pattern 1: 4 random 256-byte segments (16 lanes each)       <- what a head-major value layout would give: x0 / x0 + 1 rows adjacent
//   pattern 2: 8 random 128-byte rows (8 lanes each)            <- the current deformable-attention gather
//   pattern 3: 16 random 64-byte half rows (4 lanes each)
//   pattern 4: 64 random 16-byte pieces
// hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void __launch_bounds__(256) gather_k(const uint4* __restrict__ buf, const uint32_t* __restrict__ idx, uint4* __restrict__ out, int iters, int pattern,
                                                uint32_t mask) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
  uint32_t h = idx[gw * 64 + lane] ;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int gl = pattern == 0 ? 64 : pattern == 1 ? 16 : pattern == 2 ? 8 : pattern == 3 ? 4 : 1;      // lanes per contiguous segment
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    uint32_t seg = __shfl(h, lane - lane % gl, 64);                 // one random number per segment
    seg = (seg >> 8) & mask;                                        // segment index (units of gl * 16 bytes)
    const uint4 v = buf[(size_t)seg * gl + lane % gl];
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const size_t bytes = 8u << 20;                                  // 8 MB: lives in L2 (4 MB per XCD: partly) + Infinity Cache; also try 1 MB
  uint4 *buf, *out; uint32_t* idx;
  const int blocks = 256 * 8, iters = 2000;
  hipMalloc(&buf, 64u << 20); hipMalloc(&out, blocks * 256 * sizeof(uint4)); hipMalloc(&idx, blocks * 256 * 4);
  hipMemset(buf, 1, 64u << 20);
  std::vector<uint32_t> h(blocks * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u);
  hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t span : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
    for (int pattern = 0; pattern < 5; ++pattern) {
      const int gl = pattern == 0 ? 64 : pattern == 1 ? 16 : pattern == 2 ? 8 : pattern == 3 ? 4 : 1;
      const uint32_t nseg = (uint32_t)(span / (gl * 16));
      const uint32_t mask = nseg - 1;
      gather_k<<<blocks, 256>>>(buf, idx, out, 50, pattern, mask);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      gather_k<<<blocks, 256>>>(buf, idx, out, iters, pattern, mask);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_cu = (double)blocks * 4 * iters / 256.0;
      const double ns_per_instr = ms * 1e6 / instr_per_cu;
      printf("span %3zu MB pattern %d (%2d lanes / segment): %7.3f ms  %6.2f ns per wave-instruction and CU (= %5.1f cycles at 2.4 GHz)  %6.2f TB/s\n", span >> 20, pattern, gl, ms,
             ns_per_instr, ns_per_instr * 2.4, (double)blocks * 4 * iters * 1024 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
