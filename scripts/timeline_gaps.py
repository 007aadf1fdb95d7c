#!/usr/bin/env python
"""GPU idle gaps in a rocprofv3 --kernel-trace rocpd sqlite: the union of all kernels' [start, end) over every queue, the gaps
longer than `min_us` between busy stretches with the kernels either side, and the busy/idle totals between the first and the last
kernel whose name contains `anchor` (default: the training passes).   usage: timeline_gaps.py <db> [min_us] [anchor]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_user_pass'
    c = sqlite3.connect(db)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    src = 'kernels' if 'kernels' in views else [v for v in views if 'kernel' in v.lower()][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % src)]
    name = [x for x in cols if x in ('name', 'kernel_name', 'kernel')][0]
    start = [x for x in cols if x.lower() in ('start', 'start_timestamp', 'begin')][0]
    end = [x for x in cols if x.lower() in ('end', 'end_timestamp', 'stop')][0]
    rows = sorted(c.execute('select %s, %s, %s from %s' % (name, start, end, src)), key=lambda r: r[1])
    first = next(i for i, r in enumerate(rows) if anchor in r[0])
    last = max(i for i, r in enumerate(rows) if anchor in r[0])
    rows = rows[first:last + 1]
    t0 = rows[0][1]
    busy_end, busy, gaps, prev = rows[0][1], 0.0, [], rows[0][0]
    cur_start = rows[0][1]
    for n, s, e in rows:
        if s > busy_end:
            busy += busy_end - cur_start
            if (s - busy_end) / 1e3 >= min_us:
                gaps.append(((busy_end - t0) / 1e6, (s - busy_end) / 1e3, prev.split('(')[0][:60], n.split('(')[0][:60]))
            cur_start = s
        if e > busy_end:
            busy_end, prev = e, n
    busy += busy_end - cur_start
    total = busy_end - t0
    print('window %.2f ms, busy %.2f ms (%.1f %%), %d gaps >= %.0f us totalling %.2f ms' %
          (total / 1e6, busy / 1e6, 100.0 * busy / total, len(gaps), min_us, sum(g[1] for g in gaps) / 1e3))
    for at, us, a, b in gaps:
        print('  at %9.2f ms: idle %8.1f us   after %-50s before %s' % (at, us, a, b))


if __name__ == '__main__':
    main()
