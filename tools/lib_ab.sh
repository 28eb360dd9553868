#!/bin/bash
# The product library against A/B builds of it (tools/build_variant.sh, or an older round's library
# built from `git archive` into sporco_amd/variants/) in ONE GPU call, alternating:
#   gpurun -- 'bash tools/lib_ab.sh bench  <configs> <lib> [<lib> ...]'   bench.py values + kernel ms
#   gpurun -- 'bash tools/lib_ab.sh trace  <configs> <lib> [<lib> ...]'   rocprofv3 working averages
# <lib> = "product" or the tag of sporco_amd/variants/libsporco_amd_<tag>.so; <configs> as bench.py
# --configs (e.g. config3 or config3,config5).  Lines go to stdout; `trace` also writes
# gpurun_out/lib_ab_<configs>_<n>_<lib>.csv.  (profiles/r06_r6_vs_r5_library.md was made with it.)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
MODE=$1; CONFIGS=$2; shift 2
Q="--no-cpu-baseline --no-time-to-tol --no-parity --configs $CONFIGS"
n=0
for tag in "$@"; do
  n=$((n+1))
  lib=""; [ "$tag" != product ] && lib=$PWD/sporco_amd/variants/libsporco_amd_$tag.so
  if [ "$MODE" = trace ]; then
    rm -rf /tmp/ab$n
    (cd /tmp && SPORCO_AMD_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ab$n -o ks -- python "$OLDPWD/bench.py" $Q > /dev/null 2>&1)
    out=gpurun_out/lib_ab_${CONFIGS//,/_}_${n}_$tag.csv
    python tools/rocpd_summary.py "$(find /tmp/ab$n -name '*.db' | head -1)" $out > /dev/null 2>&1
    echo "== $tag"; grep -E "rows_|cols_|pgm_|ccmod_" $out | sort -t, -k3 -n -r | head -12 | cut -c1-250
  else
    SPORCO_AMD_LIBRARY=$lib python bench.py $Q 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
out = {'lib': '$tag', 'value': round(d['value'], 1), 'steady': round(d['steady_state']['value'], 1), 'epilogue_ms': d['roofline']['avg_kernel_ms'], 'placement': d['roofline']['placement_compact']}
for k, v in d['configs'].items():
    out[k] = round(v['value'], 2)
    out[k + '_kern'] = {n: x['avg_ms'] for n, x in v['kernels'].items() if x['avg_ms'] > 0.2}
print(json.dumps(out))"
  fi
done
