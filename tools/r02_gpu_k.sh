#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity --steps 60 --fastsolve"
for cfg in "1 0" "2 4" "4 2" "4 4" "8 1" "8 2"; do
  set -- $cfg
  SPORCO_AMD_ROWS_STAGGER_GROUPS=$1 SPORCO_AMD_ROWS_STAGGER_SLEEPS=$2 timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rows stagger $1 x $2', round(d['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})"
done
python bench.py --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default 20 steps', round(d['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})"
