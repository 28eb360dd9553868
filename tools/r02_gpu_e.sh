#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests/test_device_loop.py tests/test_parity_baseline_shapes.py tests/test_fused_xstep.py tests/test_admm_cbpdn.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
for hl in 1 0; do
  if [ $hl = 1 ]; then export SPORCO_AMD_HOST_LOOP=1; else unset SPORCO_AMD_HOST_LOOP; fi
  timeout 300 $B > $O/bench20_host$hl.json 2>$O/err
  timeout 300 $B --steps 100 > $O/bench100_host$hl.json 2>>$O/err
done
unset SPORCO_AMD_HOST_LOOP
python - <<'PY'
import json
for n in ('bench20_host1','bench20_host0','bench100_host1','bench100_host0'):
    try:
        d=json.load(open('gpurun_out/r02e/%s.json'%n)); print(n, round(d['value'],1), round(d['ms_per_step'],3), 'kernels', round(sum(d['kernels_ms_per_iter'].values()),3), 'other', round(d['other_options']['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
timeout 600 python bench.py > $O/bench_full.json 2>$O/err_full; tail -c 1500 $O/bench_full.json
