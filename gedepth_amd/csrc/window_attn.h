// Geometry shared by the VALU and MFMA window-attention kernels.
#pragma once
#include "common.h"

#define WS 7
#define WT 49          // tokens per window
#define HD 32          // head dim (every Swin stage of the reference)
#define NBIAS 169      // (2*7-1)^2
#define QLD 33         // padded row (lane reads own row: conflict-free)
#define PLD 65         // 49x49 tile, odd leading dim: conflict-free row-wise and column-wise

struct WinGeom { int B, H, W, Hp, Wp, nWh, nWw, nH, shift, C; };

// token t of window (wy,wx) -> source token index in the un-padded (H,W) grid, or -1 for a pad token;
// `region` = shift-mask region id on the padded+rolled grid (depthformer_swin.py:305-320).
__device__ __forceinline__ int win_token(const WinGeom& g, int wy, int wx, int t, int& region) {
  const int i = t / WS, j = t - i * WS;
  const int hr = wy * WS + i, wr = wx * WS + j;
  int h = hr, w = wr;
  region = 0;
  if (g.shift > 0) {
    h = hr + g.shift; if (h >= g.Hp) h -= g.Hp;
    w = wr + g.shift; if (w >= g.Wp) w -= g.Wp;
    const int rh = (hr >= g.Hp - WS) + (hr >= g.Hp - g.shift);
    const int rw = (wr >= g.Wp - WS) + (wr >= g.Wp - g.shift);
    region = 3 * rh + rw;
  }
  return (h < g.H && w < g.W) ? h * g.W + w : -1;
}

