#!/bin/bash
# Builds libgedepth_hip.so variants that differ in the -D configuration of csrc/conv1x1_bn.hip (stdin: one configuration per line) into
# tools/ubench/conv1x1_bn/bin/lib_<n>.so; `variants.sh run` times each with time.py on the GPU box.
cd "$(dirname "$0")/../../.."
D=tools/ubench/conv1x1_bn
if [ "$1" = run ]; then
  while read -r n cfg; do echo "=== $n $cfg"; GE_LIB=$D/bin/lib_$n.so timeout 120 python $D/time.py 2>&1 | grep -E "ms per|conv1x1_bn"; done < $D/bin/libs.txt
  exit 0
fi
mkdir -p $D/bin
rm -f $D/bin/lib_*.so $D/bin/libs.txt
objs=$(ls gedepth_amd/csrc/build/*.o | grep -v conv1x1_bn.o)
n=0
while IFS= read -r cfg; do
  [ -z "$cfg" ] && continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-inline-asm $cfg -c gedepth_amd/csrc/conv1x1_bn.hip -o /tmp/cb_var_$n.o 2>&1 | grep -E "error" | head -3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/cb_var_$n.o -o $D/bin/lib_$n.so ) &
  echo "$n $cfg" >> $D/bin/libs.txt
  n=$((n+1))
done
wait
ls $D/bin/
