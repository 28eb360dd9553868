#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_device_loop.py tests/test_parity_baseline_shapes.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
cd /tmp && export TMPDIR=/tmp
for hl in 0 1; do
  if [ $hl = 1 ]; then export SPORCO_AMD_HOST_LOOP=1; else unset SPORCO_AMD_HOST_LOOP; fi
  rocprofv3 --kernel-trace -d /tmp/g$hl -o ks -- $B --steps 60 --fastsolve > /tmp/g$hl.log 2>&1
  python $R/tools/rocpd_gaps.py $(find /tmp/g$hl -name "*.db" | head -1) $O/gaps_fastsolve_host$hl.csv || tail -3 /tmp/g$hl.log
  rocprofv3 --kernel-trace -d /tmp/h$hl -o ks -- $B --steps 60 > /tmp/h$hl.log 2>&1
  python $R/tools/rocpd_gaps.py $(find /tmp/h$hl -name "*.db" | head -1) $O/gaps_default_host$hl.csv || tail -3 /tmp/h$hl.log
done
head -12 $O/gaps_default_host0.csv; head -8 $O/gaps_default_host1.csv; head -8 $O/gaps_fastsolve_host0.csv
