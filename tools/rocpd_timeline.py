#!/usr/bin/env python3
"""Kernels of one period of a rocprofv3 --kernel-trace capture (rocpd .db): everything between the
`occurrence`-th and the next dispatch of the kernel whose name contains `anchor`, as
(start offset us, duration us, gap before us, name); runs of the same kernel pair are folded.

    python tools/rocpd_timeline.py <results.db> <anchor> [occurrence]
"""
import re
import sqlite3
import sys


def table(con, prefix):
    names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    return next(n for n in names if n.startswith(prefix))


def short(name):
    name = re.sub(r'void |sporco_amd::|\(anonymous namespace\)::', '', name)
    return name.split('(')[0][:44]


def main():
    con = sqlite3.connect(sys.argv[1])
    kd, ks = table(con, 'rocpd_kernel_dispatch'), table(con, 'rocpd_info_kernel_symbol')
    kcols = [r[1] for r in con.execute('pragma table_info(%s)' % ks)]
    namecol = 'display_name' if 'display_name' in kcols else 'kernel_name'
    rows = con.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id '
                       'order by d.start' % (namecol, kd, ks)).fetchall()
    anchor, occ = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    a, b = idx[occ], idx[occ + 1]
    t0 = rows[a][1]
    busy = sum(r[2] - r[1] for r in rows[a:b])
    print('period %.1f us, kernels busy %.1f us, %d dispatches' %
          ((rows[b][1] - t0) / 1e3, busy / 1e3, b - a))
    prev_end = rows[a - 1][2] if a else t0
    for n, s, e in rows[a:b]:
        print('%9.1f %8.1f %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(n)))
        prev_end = e


if __name__ == '__main__':
    main()
