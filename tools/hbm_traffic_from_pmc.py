#!/usr/bin/env python3
"""profiles/hbm_traffic_bytes.json from the two rocprofv3 --pmc passes of `python bench.py`
(FETCH_SIZE and WRITE_SIZE, separate runs, summarised by tools/rocpd_summary.py):
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE counts 64 B
per 128-B request (MI355X_MICROARCH.md, HBM section).

    python tools/hbm_traffic_from_pmc.py <pmc_fetch_size.csv> <pmc_write_size.csv> <out.json> [tag] [kernel_stats.csv]
"""
import csv
import json
import sys


def counters(path, name):
    rows = list(csv.reader(open(path)))
    start = [i for i, r in enumerate(rows) if len(r) > 1 and r[0] == 'Name' and r[1] == 'Counter'][0]
    # (the average over the dispatches that did work, when the summary has that column: the
    # device-driven solve also enqueues launches that return at once)
    return {r[0]: float(r[5] if len(r) >= 6 and r[5] else r[3]) for r in rows[start + 1:]
            if len(r) >= 4 and r[1] == name}


KERNELS = {      # bench.py's kernel names -> the leading template arguments of the dispatch name
    # (later rounds appended defaulted arguments -- the points per thread, ", 32>" -- so a name
    # matches when the arguments listed here are followed by ',' or '>')
    'rows_fwd': 'rows_fwd_kernel<16, false, false, false',
    'rows_fwd_v': 'rows_fwd_kernel<16, false, true, false',
    'fused_cols_sm': 'fused_cols_kernel<32, 16, 1, 64, false, false, false, 0',
    'rows_inv_post': 'rows_inv_post_kernel<16, false, 0, false, false, 0',
    'rows_inv_post_emit': 'rows_inv_post_kernel<16, false, 0, true, false, 0',
    'rows_inv_post_v': 'rows_inv_post_kernel<16, false, 0, false, false, 2',
    'rows_inv_post_v_emit': 'rows_inv_post_kernel<16, false, 0, true, false, 2',
}


def matches(sub, name):
    i = name.find(sub)
    return i >= 0 and name[i + len(sub):i + len(sub) + 1] in (',', '>')


def main():
    f, w = counters(sys.argv[1], 'FETCH_SIZE'), counters(sys.argv[2], 'WRITE_SIZE')
    tag = sys.argv[4] if len(sys.argv) > 4 else ''
    out = {'_comment': "HBM bytes per launch at config 2 (512x512, K=64, N=32, f32) from rocprofv3 --pmc "
                       "FETCH_SIZE / WRITE_SIZE (separate passes of `python bench.py --steps 6 --warmup 2`%s): "
                       "(2*FETCH_SIZE + WRITE_SIZE)*1024 -- gfx950 FETCH_SIZE counts 64 B per 128-B request "
                       "(MI355X_MICROARCH.md, HBM section; confirmed in round 1 on gram_kernel, which reads "
                       "the 67.4 MB Df once and reports FETCH_SIZE*1024 = 33.7 MB)." % (', ' + tag if tag else '')}
    for key, sub in KERNELS.items():
        fk = [v for n, v in f.items() if matches(sub, n)]
        wk = [v for n, v in w.items() if matches(sub, n)]
        if fk and wk:
            out[key] = (2.0 * fk[0] + wk[0]) * 1024.0
    # optional 5th argument: the kernel statistics of the --kernel-trace --stats run of the same
    # command (tools/rocpd_summary.py): the working-dispatch averages bench.py quotes beside its
    # own HIP-event timings
    if len(sys.argv) > 5:
        rows = list(csv.reader(open(sys.argv[5])))
        hdr = rows[0]
        wi = hdr.index('WorkingAverageNs')
        avg = {}
        for key, sub in KERNELS.items():
            for r in rows[1:]:
                if len(r) > wi and matches(sub, r[0]):
                    avg[key] = float(r[wi]) * 1e-6
        out['_rocprof_avg_ms'] = avg
        out['_rocprof_source'] = 'rocprofv3 --kernel-trace --stats of `python bench.py` (working dispatches)'
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
