#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "$PWD/sporco_amd/libsporco_amd_w4.so"; do
  echo "== library: ${lib:-default (2 waves per SIMD for the 8-wave kernels)}"
  export SPORCO_AMD_LIBRARY=$lib
  [ -z "$lib" ] && unset SPORCO_AMD_LIBRARY
  timeout 300 python tools/bench_other.py dl c1 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
  timeout 300 python tools/bench_dictlearn_cns.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-1000
done
