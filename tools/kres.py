#!/usr/bin/env python3
"""Kernel resource table of one .hip file: VGPRs, SGPRs, scratch, occupancy (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I.', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name = None
rows = {}
for l in out.split('\n'):
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'sporco_amd::|\(anonymous namespace\)::|void ', '', name).split('(')[0]
        rows[name] = {}
        continue
    m = re.search(r'remark:\s+(\w[\w /\[\]]*): (\d+)', l)
    if m and name: rows[name][m.group(1).strip()] = int(m.group(2))
for n, r in rows.items():
    print('%-60s VGPR %3d SGPR %3d scratch %4d occ %d' % (n[:60], r.get('VGPRs', -1), r.get('TotalSGPRs', -1), r.get('ScratchSize [bytes/lane]', -1), r.get('Occupancy [waves/SIMD]', -1)))
