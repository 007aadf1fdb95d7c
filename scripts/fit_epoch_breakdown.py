#!/usr/bin/env python
"""Per-epoch anatomy of a traced fit() (rocprofv3 --kernel-trace rocpd sqlite): an epoch = first user pass of one training call
to the first user pass of the next; for every epoch the wall time, the GPU-busy time (union over queues), and per kernel
category the summed kernel time and the time during which ONLY that category ran.   usage: fit_epoch_breakdown.py <db> [gap_us]"""
import sqlite3
import sys

CATS = [('passes', ('k_user_pass', 'k_item_pass', 'k_user_stitch', 'k_item_stitch')),
        ('sampler', ('k_mt_', 'k_accept', 'k_scan_counts', 'k_rng_finalize')),
        ('sort', ('k_rs_', 'k_item_long_flags', 'k_user_long_flags')),
        ('shuffle', ('k_fy', 'k_gather')),
        ('torch', ('at::native', 'elementwise', 'reduce_kernel')),
        ('copies', ('copyBuffer', 'fillBuffer'))]


def cat(name):
    for c, keys in CATS:
        if any(k in name for k in keys):
            return c
    return 'other'


def main():
    db = sys.argv[1]
    gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
    c = sqlite3.connect(db)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    src = 'kernels' if 'kernels' in views else [v for v in views if 'kernel' in v.lower()][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % src)]
    name = [x for x in cols if x in ('name', 'kernel_name', 'kernel')][0]
    start = [x for x in cols if x.lower() in ('start', 'start_timestamp', 'begin')][0]
    end = [x for x in cols if x.lower() in ('end', 'end_timestamp', 'stop')][0]
    rows = sorted(c.execute('select %s, %s, %s from %s' % (name, start, end, src)), key=lambda r: r[1])
    ups = [r for r in rows if 'k_user_pass' in r[0]]
    # epoch boundaries: a user pass that starts more than gap_us after the previous ITEM pass ended ... simpler: the 1st, 33rd, ...
    # user pass is not known here, so split where consecutive user passes are further apart than gap_us
    bounds = [ups[0][1]]
    for a, b in zip(ups, ups[1:]):
        if (b[1] - a[2]) / 1e3 > gap_us:
            bounds.append(b[1])
    bounds.append(rows[-1][2])
    print('%d user passes, %d epochs (split at gaps > %.0f us between user passes)' % (len(ups), len(bounds) - 1, gap_us))
    for e in range(len(bounds) - 1):
        lo, hi = bounds[e], bounds[e + 1]
        ev = []
        tot = {}
        n_up = 0
        for n, s, t in rows:
            if t <= lo or s >= hi:
                continue
            s, t = max(s, lo), min(t, hi)
            k = cat(n)
            tot[k] = tot.get(k, 0) + (t - s)
            n_up += 'k_user_pass' in n
            ev.append((s, 1, k))
            ev.append((t, -1, k))
        ev.sort()
        active, last, busy, only = {}, lo, 0, {}
        for ts, d, k in ev:
            if ts > last:
                live = [x for x, v in active.items() if v > 0]
                if live:
                    busy += ts - last
                    if len(live) == 1:
                        only[live[0]] = only.get(live[0], 0) + (ts - last)
                last = ts
            active[k] = active.get(k, 0) + d
        print('epoch %d: %.2f ms wall, busy %.2f ms, %d user passes' % (e, (hi - lo) / 1e6, busy / 1e6, n_up))
        for k in sorted(tot, key=lambda x: -tot[x]):
            print('    %-8s kernel time %7.2f ms   alone on the GPU %7.2f ms' % (k, tot[k] / 1e6, only.get(k, 0) / 1e6))


if __name__ == '__main__':
    main()
