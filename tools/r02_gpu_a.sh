#!/bin/bash
# round 2, first GPU call: new baseline-shape parity tests, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests/test_parity_baseline_shapes.py tests/test_fused_xstep.py -m gpu -q -x \
    --deselect tests/test_parity_baseline_shapes.py::test_stopping_iteration_fused_f32_vs_reference \
    > gpurun_out/r02a/pytest_parity.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest_parity.log
tail -15 gpurun_out/r02a/pytest_parity.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench.json
tail -5 gpurun_out/r02a/bench.err
