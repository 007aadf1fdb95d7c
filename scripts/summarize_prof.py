"""Summarises a rocprofv3 rocpd sqlite (`--kernel-trace --stats`) into a markdown table
under profiles/.  usage: python scripts/summarize_prof.py <db> <out.md> "<title>" [bench.json]"""
import sqlite3
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    if 'radix_sort_onesweep_iteration' in n:
        return 'rocprim radix_sort onesweep_iteration <%s>' % (
            'u32,u64' if 'unsigned long const*, unsigned long*' in n else 'u32,u32')
    if 'rocprim' in n:
        return 'rocprim ' + ('onesweep_global_offsets' if 'global_offsets' in n else 'other')
    if 'at::native' in n:
        return 'torch: ' + ('normal_ init' if 'normal' in n else 'randint ids' if 'random_from_to' in n else 'elementwise')
    return n.split('(')[0]


def main():
    db, out, title = sys.argv[1:4]
    c = sqlite3.connect(db)
    agg = {}
    for n, calls, tot, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        a = agg.setdefault(short(n), [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    with open(out, 'w') as f:
        f.write('# %s\n\nrocprofv3 --kernel-trace --stats (rocpd `top_kernels` view; long template names shortened; '
                'durations in microseconds)\n\n| kernel | calls | total us | avg us | %% |\n|---|---|---|---|---|\n' % title)
        for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('| %s | %d | %.1f | %.1f | %.1f |\n' % (k, calls, tot, tot / calls, pct))
        if len(sys.argv) > 4:
            f.write('\nbench.py line of the same build:\n\n```\n' + open(sys.argv[4]).read().strip() + '\n```\n')


if __name__ == '__main__':
    main()
