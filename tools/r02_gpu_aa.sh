#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
for cfg in "default" "2 1 0" "2 4 2" "2 8 2" "0 1 0"; do
  set -- $cfg
  if [ "$1" = default ]; then
    r=$(timeout 200 $B 2>/dev/null)
  else
    r=$(SPORCO_AMD_ROWS_PERSIST=$1 SPORCO_AMD_ROWS_STAGGER_GROUPS=$2 SPORCO_AMD_ROWS_STAGGER_SLEEPS=$3 timeout 200 $B 2>/dev/null)
  fi
  echo "$r" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rows persist/stagger $cfg:', round(d['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})"
done
