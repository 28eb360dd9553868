#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) capture: per-kernel duration statistics
(what `--stats` prints) and, when present, per-kernel averages of PMC counters.

    python tools/rocpd_summary.py <results.db> [out.csv]
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def table(con, prefix):
    names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    return next(n for n in names if n.startswith(prefix))


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    kd = table(con, 'rocpd_kernel_dispatch')
    ks = table(con, 'rocpd_info_kernel_symbol')
    cols = [r[1] for r in con.execute('pragma table_info(%s)' % kd)]
    kcols = [r[1] for r in con.execute('pragma table_info(%s)' % ks)]
    namecol = 'display_name' if 'display_name' in kcols else 'kernel_name'
    rows = con.execute('select d.id, s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id'
                       % (namecol, kd, ks)).fetchall()
    dur = defaultdict(list)
    name_of = {}
    ddur = {}
    for did, name, st, en in rows:
        dur[name].append(en - st)
        name_of[did] = name
        ddur[did] = en - st
    total = float(sum(sum(v) for v in dur.values()))
    # "Working" columns: the dispatches that lasted at least a quarter of the kernel's longest one.
    # The device-driven solve enqueues launches that return at once (the variant of the row
    # epilogue the control block did not pick, everything after the stopping iteration): they
    # count in Calls / AverageNs, not in WorkingCalls / WorkingAverageNs.
    out = [['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs',
            'WorkingCalls', 'WorkingAverageNs']]
    cut = {name: 0.25 * max(v) for name, v in dur.items()}
    for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        wv = [x for x in v if x >= cut[name]]
        out.append([name, len(v), sum(v), '%.1f' % (sum(v) / len(v)), '%.2f' % (100 * sum(v) / total),
                    min(v), max(v), len(wv), '%.1f' % (sum(wv) / len(wv))])
    # PMC counters, if any
    try:
        pe = table(con, 'rocpd_pmc_event')
        pi = table(con, 'rocpd_info_pmc')
        pcols = [r[1] for r in con.execute('pragma table_info(%s)' % pe)]
        picols = [r[1] for r in con.execute('pragma table_info(%s)' % pi)]
        pname = 'name' if 'name' in picols else 'symbol'
        q = ('select e.event_id, i.%s, e.value from %s e join %s i on e.pmc_id = i.id' % (pname, pe, pi))
        ev = table(con, 'rocpd_event')
        # event_id of a pmc row refers to the dispatch's event; map dispatch -> event
        dmap = dict(con.execute('select event_id, id from %s' % kd).fetchall()) if 'event_id' in cols else {}
        pm = defaultdict(lambda: defaultdict(list))
        pmw = defaultdict(lambda: defaultdict(list))
        for eid, cname, val in con.execute(q):
            did = dmap.get(eid)
            if did in name_of:
                pm[name_of[did]][cname].append(val)
                if ddur[did] >= cut[name_of[did]]:
                    pmw[name_of[did]][cname].append(val)
        if pm:
            out.append([])
            out.append(['Name', 'Counter', 'Dispatches', 'AveragePerDispatch', 'WorkingDispatches',
                        'AveragePerWorkingDispatch'])
            for name in pm:
                for cname, v in pm[name].items():
                    wv = pmw[name][cname]
                    out.append([name, cname, len(v), '%.1f' % (sum(v) / len(v)), len(wv),
                                '%.1f' % (sum(wv) / len(wv)) if wv else ''])
    except StopIteration:
        pass
    w = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == '__main__':
    main()
