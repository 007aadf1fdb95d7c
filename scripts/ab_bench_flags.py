#!/usr/bin/env python
"""Same-box A/B of bench.py flag sets: scripts/ab_bench_flags.py <rounds> <common flags> -- <flags A> -- <flags B> ...
Prints ms/step, the dominant kernel's roofline fraction and the per-kernel ms/step of every run (alternating A, B, A, B ...)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rounds = int(sys.argv[1])
    groups, cur = [], []
    for a in sys.argv[2:]:
        if a == '--':
            groups.append(cur)
            cur = []
        else:
            cur.append(a)
    groups.append(cur)
    common, variants = groups[0], groups[1:]
    for r in range(rounds):
        for v in variants:
            p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common + v, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               universal_newlines=True)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(json.dumps({'flags': v, 'error': p.stderr[-400:]}))
                continue
            d = json.loads(line[-1])
            roof = d['roofline']
            print(json.dumps({'flags': v, 'ms_per_step': round(d['ms_per_step'], 4), 'kernel': roof['kernel'], 'frac': round(roof['frac'], 4),
                              'kernels_ms_per_step': {k: round(x.get('avg_ms', 0), 4) for k, x in roof.get('kernels', {}).items()}
                              if isinstance(roof.get('kernels'), dict) else None}), flush=True)


if __name__ == '__main__':
    main()
