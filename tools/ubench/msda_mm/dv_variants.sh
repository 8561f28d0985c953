#!/bin/bash
# Builds libgedepth_hip.so variants that differ in the -D configuration of csrc/msda_mm.hip (stdin: one configuration per line) into
# tools/ubench/msda_mm/bin/lib_<n>.so (+ bin/libs.txt); `dv_variants.sh run [args]` times each with dv_time.py on the GPU box.
cd "$(dirname "$0")/../../.."
if [ "$1" = run ]; then
  shift
  while read -r n cfg; do echo "=== $n $cfg"; GE_LIB=tools/ubench/msda_mm/bin/lib_$n.so python tools/ubench/msda_mm/dv_time.py "$@" 2>&1 | grep -E "us$"; done < tools/ubench/msda_mm/bin/libs.txt
  exit 0
fi
mkdir -p tools/ubench/msda_mm/bin
rm -f tools/ubench/msda_mm/bin/lib_*.so tools/ubench/msda_mm/bin/libs.txt
objs=$(ls gedepth_amd/csrc/build/*.o | grep -v msda_mm.o)
n=0
while IFS= read -r cfg; do
  [ -z "$cfg" ] && continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-inline-asm $cfg -c gedepth_amd/csrc/msda_mm.hip -o /tmp/mm_var_$n.o 2>&1 | grep -E "error|failed to meet" | head -3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mm_var_$n.o -o tools/ubench/msda_mm/bin/lib_$n.so ) &
  echo "$n $cfg" >> tools/ubench/msda_mm/bin/libs.txt
  n=$((n+1))
done
wait
ls -la tools/ubench/msda_mm/bin/
