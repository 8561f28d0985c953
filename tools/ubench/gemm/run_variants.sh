#!/bin/bash
# A/B of csrc/gemm.hip build variants (tools/ubench/gemm/bin/gemm_<name>.so, built by build_variants.sh) on the large token shapes.
for f in tools/ubench/gemm/bin/gemm_*.so; do echo "== $f"; python tools/ubench/gemm_time.py --lib $f --no-lib --only "${ONLY:-neck cross,s3 fc2,s0 fc1}" 2>&1 | grep -v amdgpu.ids; done
