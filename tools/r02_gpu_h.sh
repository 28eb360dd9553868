#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02h; mkdir -p $O
timeout 1200 python -m pytest tests/test_fused_xstep.py tests/test_parity_baseline_shapes.py tests/test_gpu_fullsize.py tests/test_device_loop.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/bench_config3.py > $O/config3.jsonl 2>$O/config3.err; cat $O/config3.jsonl | tail -5; tail -3 $O/config3.err
SPORCO_AMD_JOINT_SEPARATE=1 timeout 600 python tools/bench_config3.py > $O/config3_separate.jsonl 2>>$O/config3.err; tail -2 $O/config3_separate.jsonl
