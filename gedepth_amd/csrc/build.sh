#!/bin/bash
# Build libgedepth_hip.so (gfx950 only) in-tree.  Usage: gedepth_amd/csrc/build.sh
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-inline-asm \
  aug.hip conv3x3.hip conv3x3_c1.hip conv3x3_wgrad.hip conv1x1_wgrad.hip decoder.hip gemm.hip ground.hip msda.hip msda_drain_mfma.hip msda_mm.hip msda_win.hip neck.hip nhwc.hip norm.hip window_attn.hip window_attn_mfma.hip -o libgedepth_hip.so "$@"
echo "built $(pwd)/libgedepth_hip.so"
