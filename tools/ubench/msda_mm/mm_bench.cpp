// Standalone timing harness for csrc/msda_mm.hip (round 4 bring-up): bench-shape inputs (8 x 352 x 1120: 4 value levels, cross-attention
// queries 176 x 560 with the bench model's reference points, self-attention queries = the levels), kernel variants chosen with -D macros.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics [-DMM_...] tools/ubench/msda_mm/mm_bench.cpp -o /tmp/mm_bench && /tmp/mm_bench
#include "../../../gedepth_amd/csrc/msda_mm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <string>

static std::vector<char> slurp(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { fprintf(stderr, "missing %s\n", p.c_str()); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> v(n); if (fread(v.data(), 1, n, f) != (size_t)n) exit(3); fclose(f); return v;
}
static uint16_t h_bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tools/ubench/msda_mm/data/";
  const int B = 8, nH = 8, L = 4, P = 8;
  const int hw[8] = {88, 280, 44, 140, 22, 70, 11, 35};
  int Nv = 0; for (int l = 0; l < 4; ++l) Nv += hw[2 * l] * hw[2 * l + 1];
  auto bias = slurp(dir + "offset_bias.bin");
  const float* bs = (const float*)bias.data();
  // cheap noise (sum of three uniforms, unit variance): the inputs only need the right statistics
  uint64_t st = 88172645463325252ull;
  auto nd = [&](int) {
    float a = 0.f;
    for (int i = 0; i < 3; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += (float)(st >> 40) * (1.f / 16777216.f) - 0.5f; }
    return 2.f * a;
  };
  int rng = 0;
  std::vector<uint16_t> hv((size_t)B * Nv * 512);
  for (auto& x : hv) x = h_bf(nd(rng));
  bf16_t* d_value; CK(hipMalloc(&d_value, hv.size() * 2)); CK(hipMemcpy(d_value, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
  for (int which = 0; which < 2; ++which) {
    const char* name = which ? "self" : "cross";
    auto refb = slurp(dir + (which ? "ref_self.bin" : "ref_cross.bin"));
    auto ordb = slurp(dir + (which ? "order_self.bin" : "order_cross.bin"));
    const int Nq = (int)(ordb.size() / 4);
    std::vector<uint16_t> hraw((size_t)B * Nq * 768);
    for (size_t r = 0; r < (size_t)B * Nq; ++r) {
      for (int c = 0; c < 512; ++c) hraw[r * 768 + c] = h_bf(bs[c] + 0.05f * nd(rng));
      for (int c = 512; c < 768; ++c) hraw[r * 768 + c] = h_bf(0.1f * nd(rng));
    }
    bf16_t *d_raw, *d_out, *d_draw; float* d_ref; int* d_ord;
    CK(hipMalloc(&d_draw, hraw.size() * 2));
    CK(hipMalloc(&d_raw, hraw.size() * 2)); CK(hipMemcpy(d_raw, hraw.data(), hraw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)B * Nq * 512 * 2));
    CK(hipMalloc(&d_ref, refb.size())); CK(hipMemcpy(d_ref, refb.data(), refb.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ord, ordb.size())); CK(hipMemcpy(d_ord, ordb.data(), ordb.size(), hipMemcpyHostToDevice));
    for (int sorted = 1; sorted >= 0; --sorted) {
      auto run = [&]() {
        int e = ge_msda_fwd_mm(d_value, hw, d_raw, 768, d_raw + 512, 768, d_ref, 0, 2, 0, sorted ? d_ord : nullptr, nullptr, nullptr, d_out, B, Nv, Nq, nH, L, P,
                               GE_BF16, nullptr);
        if (e) { fprintf(stderr, "ge_msda_fwd_mm -> %d\n", e); exit(1); }
      };
      for (int i = 0; i < 3; ++i) run();
      CK(hipDeviceSynchronize());
      hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
      CK(hipEventRecord(s, nullptr));
      for (int i = 0; i < 10; ++i) run();
      CK(hipEventRecord(e, nullptr)); CK(hipEventSynchronize(e));
      float ms; CK(hipEventElapsedTime(&ms, s, e));
      std::vector<uint16_t> ho(4096); CK(hipMemcpy(ho.data(), d_out, 8192, hipMemcpyDeviceToHost));
      double cs = 0; for (auto x : ho) { uint32_t u = (uint32_t)x << 16; float f; memcpy(&f, &u, 4); cs += f; }
      printf("%-6s %-9s fwd %8.3f ms   (checksum %.4f)", name, sorted ? "ordered" : "raster", ms / 10, cs);
      // backward d_loc / d_attw: the forward's output doubles as the gradient
      auto runb = [&]() {
        int e2 = ge_msda_bwd_lw_mm(d_value, hw, d_raw, 768, d_raw + 512, 768, d_ref, 0, 2, 0, sorted ? d_ord : nullptr, d_out, d_draw, 768, d_draw + 512, 768,
                                   B, Nv, Nq, nH, L, P, GE_BF16, nullptr);
        if (e2) { fprintf(stderr, "ge_msda_bwd_lw_mm -> %d\n", e2); exit(1); }
      };
      for (int i = 0; i < 3; ++i) runb();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(s, nullptr));
      for (int i = 0; i < 10; ++i) runb();
      CK(hipEventRecord(e, nullptr)); CK(hipEventSynchronize(e));
      CK(hipEventElapsedTime(&ms, s, e));
      CK(hipMemcpy(ho.data(), d_draw, 8192, hipMemcpyDeviceToHost));
      cs = 0; for (auto x : ho) { uint32_t u = (uint32_t)x << 16; float f; memcpy(&f, &u, 4); cs += f; }
      printf("   bwd_lw %8.3f ms   (checksum %.4f)\n", ms / 10, cs);
    }
    hipFree(d_draw); hipFree(d_raw); hipFree(d_out); hipFree(d_ref); hipFree(d_ord);
  }
  return 0;
}
