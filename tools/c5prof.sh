cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-time-to-tol --no-parity --configs config3"
for tag in r6 slabold r5 r6; do
  lib=""; [ $tag != r6 ] && lib=$PWD/sporco_amd/variants/libsporco_amd_$tag.so
  rm -rf /tmp/c3$tag; (cd /tmp && SPORCO_AMD_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c3$tag -o ks -- python $GRAFT_REPO_ROOT/bench.py $Q > /dev/null 2>&1)
  python tools/rocpd_summary.py "$(find /tmp/c3$tag -name '*.db' | head -1)" gpurun_out/r06s_c3_$tag.csv > /dev/null 2>&1
  echo "== $tag"; grep -E "rows_inv_post_kernel<16, false, 0, (true|false), true, 2|cols_slab|rows_fwd_kernel<16, false, true, true" gpurun_out/r06s_c3_$tag.csv | cut -c1-250
done
