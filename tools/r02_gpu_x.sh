#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_pgm.py tests/test_pgm_cbpdn.py tests/test_parity_baseline_shapes.py -m gpu -x -q -k "pgm or PGM or fista or config4 or backtrack" 2>&1 | tail -4
timeout 300 python tools/bench_pgm_k128.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02x_pgm_k128.jsonl
timeout 300 python tools/bench_other.py pgm 2>&1 | grep -v amdgpu.ids
