#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace capture (rocpd .db):
for every (previous kernel -> next kernel) pair, the average of next.start - prev.end.

    python tools/rocpd_gaps.py <results.db> [out.csv]
"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def table(con, prefix):
    names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    return next(n for n in names if n.startswith(prefix))


def short(name):
    name = re.sub(r'void |sporco_amd::|\(anonymous namespace\)::', '', name)
    return name.split('(')[0][:48]


def main():
    con = sqlite3.connect(sys.argv[1])
    kd, ks = table(con, 'rocpd_kernel_dispatch'), table(con, 'rocpd_info_kernel_symbol')
    kcols = [r[1] for r in con.execute('pragma table_info(%s)' % ks)]
    namecol = 'display_name' if 'display_name' in kcols else 'kernel_name'
    rows = con.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id '
                       'order by d.start' % (namecol, kd, ks)).fetchall()
    gaps = defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        gaps[(short(n0), short(n1))].append(s1 - e0)
    out = [['Previous', 'Next', 'Count', 'AverageGapNs', 'MedianGapNs', 'MinNs', 'MaxNs']]
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        v = sorted(v)
        out.append([a, b, len(v), '%.0f' % (sum(v) / len(v)), v[len(v) // 2], v[0], v[-1]])
    w = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == '__main__':
    main()
