"""Per-kernel summary of every rocprofv3 database under a directory: durations (kernel trace) and / or mean counter values
per launch (--pmc passes).  usage: python scripts/summarize_any.py <dir> [name-width]"""
import os
import re
import sqlite3
import sys


def short(n, w):
    n = re.sub(r'\(anonymous namespace\)::', '', n.replace('void ', ''))
    n = re.sub(r'\(.*$', '', n)
    return n[:w]


def main():
    src = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    for r, _, fs in sorted(os.walk(src)):
        for f in sorted(fs):
            if not f.endswith('.db'):
                continue
            c = sqlite3.connect(os.path.join(r, f))
            tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
            print('## %s' % os.path.relpath(os.path.join(r, f), src))
            if 'top_kernels' in tabs:
                rows = list(c.execute('select name,total_calls,total_duration,average from top_kernels'))
                if rows:
                    print('| kernel | calls | avg us | total us |\n|---|---|---|---|')
                    for n, calls, tot, avg in sorted(rows, key=lambda x: -x[2])[:40]:
                        print('| %s | %d | %.2f | %.1f |' % (short(n, w), calls, avg, tot))
            if 'counters_collection' in tabs:
                q = 'select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name'
                rows = list(c.execute(q))
                if rows:
                    print('| kernel | counter | mean per launch | launches |\n|---|---|---|---|')
                    for n, cn, tot, k in sorted(rows):
                        print('| %s | %s | %.6g | %d |' % (short(n, w), cn, tot / max(k, 1), k))


if __name__ == '__main__':
    main()
