#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
for rp in 0 1; do
  if [ $rp = 0 ]; then export SPORCO_AMD_ROWS_PERSIST=0; else unset SPORCO_AMD_ROWS_PERSIST; fi
  for st in 20 100; do
  timeout 300 $B --steps $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rows_persist=$rp steps=$st', round(d['value'],1), 'fast', round(d['other_options']['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})"
  done
done
unset SPORCO_AMD_ROWS_PERSIST
timeout 900 python -m pytest tests/test_fused_xstep.py tests/test_parity_baseline_shapes.py tests/test_device_loop.py -m gpu -q -x 2>&1 | tail -3
