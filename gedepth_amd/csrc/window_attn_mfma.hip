// bf16 MFMA tile variant of the window-attention core (placeholder until implemented).
#include "common.h"
#include "window_attn.h"
extern const int ge_window_attn_mfma_available = 0;
int ge_window_attn_fwd_mfma(const void*, const float*, const float*, void*, const WinGeom&, float, hipStream_t) { return GE_ERR_UNSUPPORTED; }
int ge_window_attn_bwd_mfma(const void*, const float*, const float*, const void*, void*, float*, const WinGeom&, float, int, hipStream_t) { return GE_ERR_UNSUPPORTED; }
