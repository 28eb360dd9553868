#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo default; timeout 300 python tools/bench_ams.py 2>&1 | grep "^{" | cut -c1-330
echo no tail; SPORCO_AMD_NO_TAIL=1 timeout 300 python tools/bench_ams.py 2>&1 | grep "^{" | cut -c1-330
echo grdmsk default; timeout 300 python tools/bench_grdmsk.py 2>&1 | grep "^{" | cut -c1-200
echo grdmsk no tail; SPORCO_AMD_NO_TAIL=1 timeout 300 python tools/bench_grdmsk.py 2>&1 | grep "^{" | cut -c1-200
