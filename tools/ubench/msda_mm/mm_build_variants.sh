#!/bin/bash
# compiles tools/ubench/msda_mm/mm_bench.cpp once per -D configuration (stdin, one per line) into tools/ubench/msda_mm/bin/mm_<n>; writes bin/list.txt
cd "$(dirname "$0")/../../.."
rm -f tools/ubench/msda_mm/bin/mm_* tools/ubench/msda_mm/bin/list.txt
n=0
while IFS= read -r cfg; do
  [ -z "$cfg" ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics $cfg tools/ubench/msda_mm/mm_bench.cpp -o tools/ubench/msda_mm/bin/mm_$n 2>&1 | grep -E "error|failed to meet" | head -3 &
  echo "$n $cfg" >> tools/ubench/msda_mm/bin/list.txt
  n=$((n+1))
done
wait
