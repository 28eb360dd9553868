#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p $R/gpurun_out/r03q
cat > /tmp/g.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
os.environ['SPORCO_AMD_UNFUSED'] = '1'
os.environ['SPORCO_AMD_NO_C2R_POST'] = '1'
import numpy as np
from sporco_amd.admm import cbpdn as ac
rng = np.random.RandomState(1)
D = rng.randn(8, 8, 64).astype(np.float32); S = rng.randn(512, 512, 8).astype(np.float32)
b = ac.ConvBPDN(D, S, 0.05, ac.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
b.solve(); b._dev.sync()
b.opt['MaxMainIter'] = 20
t0 = time.perf_counter(); b.solve(); b._dev.sync(); print('it/s', 20 / (time.perf_counter() - t0))
import cProfile, pstats
b.opt['MaxMainIter'] = 20
pr = cProfile.Profile(); pr.enable(); b.solve(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/pg -o g -- python /tmp/g.py > $R/gpurun_out/r03q/generic_trace_stdout.txt 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/pg -name "*.db" | head -1) $R/gpurun_out/r03q/generic_kernel_stats.csv > /dev/null 2>&1
head -14 $R/gpurun_out/r03q/generic_kernel_stats.csv | cut -c1-150
grep -v amdgpu.ids $R/gpurun_out/r03q/generic_trace_stdout.txt | tail -40 | cut -c1-150
