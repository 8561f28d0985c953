// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which (lane, element) of the result comes from which
// (lane, element) of the plain 8-byte read at the same per-lane addresses.  Prints the map for a linear address pattern
// (lane i -> byte 8 i) and for the [4 rows][16 cols] block pattern the MFMA B operand of the deformable-attention drain uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* addr_bytes, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int a = addr_bytes[threadIdx.x];
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)((__attribute__((address_space(3))) char*)lds + a));
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(u >> (16 * j));
}

int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 2; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 8;                     // linear: lane l holds u16 elements 4l .. 4l+3
      else {                                               // rows of 64 u16 (128 B): group g = l/16 reads rows 4g'..: row = (l%16)/4 + 4*(g>>1), cols 16*(g&1) + 4*(l%4)
        const int g = l / 16, i = l % 16;
        const int row = i / 4 + 4 * (g >> 1), col = 16 * (g & 1) + 4 * (i % 4);
        h_addr[l] = (row * 64 + col) * 2;
      }
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d (element index = row*64+col for pattern 1)\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d addr %4d : %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
