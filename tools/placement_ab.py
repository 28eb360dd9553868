"""The headline workload with and without the library's placement search (api_placement.inc),
each in a fresh process, with 0 ... 24 GiB of device memory taken beforehand so that the arrays of
the run land in different physical memory (what the allocation history of a real process does).
One JSON line per run: avg launch time of the emitting row epilogue, steady iterations/s.

    gpurun -- 'bash tools/gpu.sh r05x py:placement_ab.py'
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ['--no-cpu-baseline', '--no-time-to-tol', '--no-parity', '--configs', 'none']
dummies = [int(x) for x in sys.argv[1:]] or [0, 2050, 4100, 6150, 8200, 12300, 16400, 24600, 2, 258]

for mb in dummies:
    for on in ('0', '1'):
        env = dict(os.environ, SPORCO_AMD_PLACEMENT=on)
        if mb:
            env['SPORCO_AMD_BENCH_PREALLOC_MB'] = str(mb)
        r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + QUICK, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        try:
            d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001
            print(json.dumps({'prealloc_MiB': mb, 'placement': on, 'error': str(e), 'stderr': r.stderr.decode()[-120:]}))
            continue
        rf = d['roofline']
        print(json.dumps({'prealloc_MiB': mb, 'placement_search': on == '1', 'value': round(d['value'], 1),
                          'steady': round(d['steady_state']['value'], 1), 'kernel': rf['kernel'],
                          'avg_kernel_ms': rf['avg_kernel_ms'], 'frac': round(rf['frac'], 3),
                          'decisions': [(p['role'], p['candidates'], p['same_region_GBps'], p['first_ratio'], p['chosen_ratio'])
                                        for p in rf.get('placement') or []]}), flush=True)
