// Standalone timing / self-consistency harness for csrc/conv3x3.hip at the bench shapes (8 x 352 x 1120 training step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCV_PARK_KERNEL] tools/ubench/conv3x3/conv_bench.cpp -o conv_bench && ./conv_bench
// Prints time, TFLOP/s and a checksum per shape: the round-3 kernel (-DCV_PARK_KERNEL) and the LDS-DMA kernel must print the same checksums.
#include "../../../gedepth_amd/csrc/conv3x3.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t h_bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
int main() {
  struct Shape { int N, H, W, Ci, Co; };
  const Shape shapes[] = {{8, 176, 560, 576, 64}, {8, 176, 560, 64, 576}, {8, 176, 560, 160, 64}, {8, 176, 560, 64, 64}, {8, 88, 280, 608, 96},
                          {8, 88, 280, 288, 96}, {8, 44, 140, 704, 192}, {8, 22, 70, 1152, 384}, {8, 11, 35, 1280, 768}};
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)(st >> 40) * (1.f / 16777216.f) - 0.5f; };
  for (const Shape& sh : shapes) {
    const size_t nx = (size_t)sh.N * sh.H * sh.W * sh.Ci, nw = (size_t)sh.Co * 9 * sh.Ci, ny = (size_t)sh.N * sh.H * sh.W * sh.Co;
    std::vector<uint16_t> hx(nx), hw(nw);
    for (auto& v : hx) v = h_bf(rnd());
    for (auto& v : hw) v = h_bf(rnd() * 0.1f);
    std::vector<float> hb(sh.Co);
    for (auto& v : hb) v = rnd();
    bf16_t *dx, *dw, *dy; float* db;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&db, sh.Co * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), sh.Co * 4, hipMemcpyHostToDevice));
    auto run = [&]() {
      int e = ge_conv3x3_nhwc_fwd(dx, dw, db, dy, sh.N, sh.H, sh.W, sh.Ci, sh.Co, 1, 0.01f, GE_BF16, nullptr);
      if (e) { fprintf(stderr, "ge_conv3x3_nhwc_fwd -> %d\n", e); exit(1); }
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    CK(hipEventRecord(s, nullptr));
    for (int i = 0; i < 10; ++i) run();
    CK(hipEventRecord(e, nullptr)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e)); ms /= 10;
    std::vector<uint16_t> hy(ny); CK(hipMemcpy(hy.data(), dy, ny * 2, hipMemcpyDeviceToHost));
    double cs = 0, ca = 0; for (size_t i = 0; i < ny; ++i) { uint32_t u = (uint32_t)hy[i] << 16; float f; memcpy(&f, &u, 4); cs += f; ca += f < 0 ? -f : f; }
    const double fl = 2.0 * sh.N * sh.H * sh.W * (double)sh.Ci * sh.Co * 9;
    printf("%dx%4d->%4d @%3dx%3d  %8.1f us  %7.1f TFLOP/s (%.1f %% of 2500)   sum %.6e abs %.6e\n", sh.N, sh.Ci, sh.Co, sh.H, sh.W, ms * 1e3, fl / ms * 1e-9,
           fl / ms * 1e-9 / 25.0, cs, ca);
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(db);
  }
  return 0;
}
