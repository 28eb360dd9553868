#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for t in bench_gradreg.py bench_ams.py bench_grdmsk.py bench_mcdict.py bench_joint.py bench_dictlearn_rgb.py; do
  timeout 300 python tools/$t 2>&1 | grep -v amdgpu.ids | grep "^{" | cut -c1-260
done | tee gpurun_out/r02u_side_benches.jsonl
