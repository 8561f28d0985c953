// Exhaustive host-side check for gedepth_amd/csrc/msda_mm.hip: off / W formed as q0 = off * RN(1/W), q = fma(fma(-q0, W, off), RN(1/W), q0) is
// bit-identical to IEEE division for every bf16-valued offset and every integer map size W <= 8191 (the kernel's envelope).
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
int main() {
  long bad = 0, n = 0;
  for (int W = 1; W <= 8191; ++W) {
    const float fW = (float)W;
    const float rW = 1.0f / fW;               /* correctly rounded */
    for (uint32_t h = 0; h < 65536; ++h) {
      uint32_t u = h << 16; float ox; memcpy(&ox, &u, 4);
      if (isnan(ox) || isinf(ox)) continue;
      const float ref = ox / fW;
      const float q0 = ox * rW;
      const float e = fmaf(-q0, fW, ox);
      const float q = fmaf(e, rW, q0);
      ++n;
      if (memcmp(&q, &ref, 4) != 0 && !(q == 0.f && ref == 0.f)) { if (bad < 10) printf("W %d ox %a: %a vs %a\n", W, ox, q, ref); ++bad; }
    }
  }
  printf("checked %ld, mismatches %ld\n", n, bad);
  return 0;
}
