#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 2048 1024 512 4096; do echo "blocks $b"; SPORCO_AMD_CG_BLOCKS=$b timeout 300 python tools/bench_cg_only.py 2>&1 | grep -v amdgpu.ids; done
