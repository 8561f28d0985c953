#!/bin/bash
# usage: build_variants.sh name:-DFLAG=.. [name:-DFLAG=.. ...]   ->  tools/ubench/gemm/bin/gemm_<name>.so (csrc/gemm.hip alone)
cd "$(dirname "$0")/../../../gedepth_amd/csrc"; mkdir -p ../../tools/ubench/gemm/bin
for v in "$@"; do n=${v%%:*}; f=${v#*:}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${f//,/ } gemm.hip -o ../../tools/ubench/gemm/bin/gemm_$n.so || exit 1; done
