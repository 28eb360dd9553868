#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_xstep.py tests/test_parity_baseline_shapes.py -m gpu -q -x -k "joint or config3" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
N=32 timeout 600 python tools/bench_config3.py 2>/dev/null | head -1 > $O/config3_emit.json; cat $O/config3_emit.json
SPORCO_AMD_NO_SPECULATION=1 timeout 600 python tools/bench_config3.py 2>/dev/null | head -1 > $O/config3_noemit.json; cat $O/config3_noemit.json
