#!/bin/bash
# Build libgedepth_hip.so (gfx950 only) in-tree.  Usage: gedepth_amd/csrc/build.sh [extra hipcc flags]
# One object per source, compiled in parallel (JOBS, default: all cores), only when the source or a header is newer than the object.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-inline-asm $*"
SRCS="aug.hip conv3x3.hip conv3x3_c1.hip conv3x3_wgrad.hip conv1x1_wgrad.hip conv1x1_bn.hip decoder.hip gemm.hip ground.hip msda.hip msda_drain_mfma.hip msda_mm.hip msda_win.hip neck.hip nhwc.hip norm.hip window_attn.hip window_attn_mfma.hip"
mkdir -p build
echo "$FLAGS" > build/.flags.new
if ! cmp -s build/.flags.new build/.flags 2>/dev/null; then rm -f build/*.o; mv build/.flags.new build/.flags; else rm -f build/.flags.new; fi
newest_header=$(ls -t *.h ../../include/*.h | head -1)
todo=""
for f in $SRCS; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_header" -nt "$o" ]; then todo="$todo $f"; fi
done
if [ -n "$todo" ]; then
  printf '%s\n' $todo | xargs -P "${JOBS:-$(nproc)}" -I{} sh -c "rm -f build/\$(basename {} .hip).o; $HIPCC $FLAGS -c {} -o build/\$(basename {} .hip).o 2>&1 | grep -v 'argument unused' || true; test -f build/\$(basename {} .hip).o"
fi
objs=""
for f in $SRCS; do objs="$objs build/${f%.hip}.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o libgedepth_hip.so
echo "built $(pwd)/libgedepth_hip.so"
