#!/usr/bin/env python
"""One fit() of the C1-shape model under `rocprofv3 --kernel-trace` -> where an epoch's 3+ ms go: GPU busy time per kernel
family and the idle gaps between kernels (host-side latency: launches, read-backs, torch ops).
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o c1 -- python scripts/trace_c1_fit.py run
                 python scripts/trace_c1_fit.py summarize OUT/.../c1_kernel_trace.csv"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from spotlight_amd.cross_validation import random_train_test_split
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    rs = np.random.RandomState(42)
    inter = Interactions(rs.randint(0, 943, 100000).astype(np.int32), rs.randint(0, 1682, 100000).astype(np.int32),
                         num_users=943, num_items=1682)
    train, _ = random_train_test_split(inter, random_state=np.random.RandomState(42))
    opt = os.environ.get('C1_OPT', 'adagrad')
    kw = dict(optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2)) if opt == 'adagrad' else dict(l2=1e-6)
    mk = lambda: ImplicitFactorizationModel(loss='bpr', embedding_dim=32, batch_size=1024, n_iter=10, use_cuda=True,
                                            random_state=np.random.RandomState(42), **kw)
    mk().fit(train)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    mk().fit(train)
    torch.cuda.synchronize()
    print('fit_s', time.perf_counter() - t0)


def family(name):
    for key, fam in (('k_fy', 'shuffle'), ('k_gather_rows', 'shuffle'), ('k_mt_', 'sampler'), ('k_accept', 'sampler'),
                     ('k_rng', 'sampler'), ('radix', 'sorts'), ('rocprim', 'sorts'), ('k_build', 'sorts'), ('k_bilinear_epoch', 'train'),
                     ('k_user_pass', 'train'), ('k_item_pass', 'train'), ('k_dense_sweep', 'train'), ('at::', 'torch'),
                     ('copyBuffer', 'copies'), ('fillBuffer', 'memsets'), ('k_i64_to_u32', 'sampler'), ('k_u32_to_i64', 'sampler')):
        if key in name:
            return fam
    return 'other'


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
    # the second fit = the last half of the events by time: take events after the largest idle gap in the middle third
    t_first, t_last = ev[0][0], ev[-1][1]
    fam_busy, n = {}, {}
    busy = 0
    gaps = []
    prev_end = None
    half = [e for e in ev if e[0] >= t_first + (t_last - t_first) // 2]
    for s, e, name in half:
        f = family(name)
        fam_busy[f] = fam_busy.get(f, 0) + (e - s)
        n[f] = n.get(f, 0) + 1
        busy += e - s
        if prev_end is not None and s > prev_end:
            gaps.append(s - prev_end)
        prev_end = max(prev_end or e, e)
    span = half[-1][1] - half[0][0]
    gaps.sort(reverse=True)
    out = {'window_ms': span / 1e6, 'gpu_busy_ms': busy / 1e6, 'idle_ms': sum(gaps) / 1e6, 'kernels': len(half),
           'busy_ms_by_family': {k: round(v / 1e6, 3) for k, v in sorted(fam_busy.items(), key=lambda kv: -kv[1])},
           'launches_by_family': n, 'largest_gaps_us': [round(g / 1e3, 1) for g in gaps[:12]],
           'gaps_over_20us': sum(1 for g in gaps if g > 20000), 'gaps_5_to_20us': sum(1 for g in gaps if 5000 < g <= 20000)}
    print(json.dumps(out))


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else summarize(sys.argv[2])
