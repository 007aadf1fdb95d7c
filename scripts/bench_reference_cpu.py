#!/usr/bin/env python
"""The LIVE reference (spotlight from /root/reference, CPU PyTorch) timed on the same kind of workload as bench.py,
at a size its host loop can finish: the number BASELINE.md has no published value for.  Build container only
(the reference does not exist on the GPU box).   usage: bench_reference_cpu.py [users] [items] [interactions] [batch]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.reference_import import import_reference  # noqa: E402
import_reference()  # the reference itself, not this repository's `spotlight` alias package
from spotlight.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight.interactions import Interactions  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
I = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
B = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
rs = np.random.RandomState(0)
inter = Interactions(rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32), num_users=U, num_items=I)
out = {'users': U, 'items': I, 'interactions': N, 'batch': B, 'dim': 64, 'loss': 'bpr', 'threads': torch.get_num_threads(),
       'host': 'build container (%d cores)' % os.cpu_count()}
for name, kw in (('sparse_adagrad', dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2))),
                 ('default_dense_adam', dict())):
    n = N if name == 'sparse_adagrad' else min(N, 4 * B)  # the dense default sweeps every table row each minibatch
    sub = Interactions(inter.user_ids[:n], inter.item_ids[:n], num_users=U, num_items=I)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=B, random_state=np.random.RandomState(1), **kw)
    model.fit(sub)  # warm-up epoch (initialisation, allocator)
    t0 = time.perf_counter()
    model.fit(sub)
    dt = time.perf_counter() - t0
    out[name] = {'interactions': n, 'seconds': dt, 'interactions_per_s': n / dt}
print(json.dumps(out))
