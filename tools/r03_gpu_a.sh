#!/bin/bash
# Round 3, GPU call A: the single-array state (V form) and the persistent FISTA column kernels on
# the device -- parity tests, bench A/B, config 4 A/B, rocprofv3 kernel statistics, MALL probe.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
( cd tools/ubench && timeout 120 /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o /tmp/mall_probe 2>/dev/null && timeout 120 /tmp/mall_probe > $O/mall_probe.jsonl 2> $O/mall_probe.err ) &
timeout 900 python -m pytest tests/test_vform.py tests/test_device_loop.py tests/test_fused_pgm.py tests/test_parity_baseline_shapes.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_subset.txt
wait
B="python bench.py --no-cpu-baseline --no-time-to-tol"
timeout 300 $B > $O/bench_vform.json 2> $O/bench_vform.err
SPORCO_AMD_NO_VFORM=1 timeout 300 $B --no-parity > $O/bench_yuform.json 2> $O/bench_yuform.err
timeout 300 $B --no-parity --fastsolve > $O/bench_vform_fast.json 2>/dev/null
for p in 3 0 1 2; do
  SPORCO_AMD_PGM_PERSIST=$p timeout 200 python tools/bench_other.py pgm 2>/dev/null | grep "^{" | sed "s/^{/{\"PGM_PERSIST\": $p, /" >> $O/config4.jsonl
done
for sg in "1 0" "4 4" "8 2"; do
  set -- $sg
  SPORCO_AMD_PGM_STAGGER_GROUPS=$1 SPORCO_AMD_PGM_STAGGER_SLEEPS=$2 timeout 200 python tools/bench_other.py pgm 2>/dev/null | grep "^{" | head -1 | sed "s/^{/{\"stagger\": \"$1x$2\", /" >> $O/config4.jsonl
done
timeout 200 python tools/bench_other.py dl c1 2>/dev/null | grep "^{" > $O/other.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity --steps 40 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/kernel_stats_vform.csv > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o ks -- python $R/tools/bench_other.py pgm > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/kernel_stats_config4.csv > /dev/null 2>&1
cd $R
ls -la $O; head -c 600 $O/bench_vform.json; echo; cat $O/config4.jsonl; cat $O/other.jsonl; tail -3 $O/mall_probe.jsonl
