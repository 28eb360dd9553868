#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity --steps 40 --fastsolve"
for cfg in "1 0" "2 2" "2 4" "2 6" "4 1" "4 2" "4 3" "8 1" "3 3"; do
  set -- $cfg
  SPORCO_AMD_COLS_STAGGER_GROUPS=$1 SPORCO_AMD_COLS_STAGGER_SLEEPS=$2 timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('groups $1 sleeps $2', round(d['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})"
done
