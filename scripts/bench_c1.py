#!/usr/bin/env python
"""BASELINE.json configs[0] (C1), on the MovieLens-100K SHAPE (the dataset itself cannot be downloaded here):
synthetic uniform ids, 943 users x 1682 items, 100 000 interactions, 80/20 split, ImplicitFactorizationModel(
loss='bpr', embedding_dim=32, batch_size=1024, n_iter=10, learning_rate=1e-2, l2=1e-6) with the reference's
DEFAULT optimizer (dense Adam + l2: every row of every table is updated every step) -- the setting of
tests/factorization/test_implicit.py:40-57 of the reference.  Prints end-to-end fit() time and interactions/s."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.cross_validation import random_train_test_split  # noqa: E402
from spotlight_amd.evaluation import mrr_score  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402

rs = np.random.RandomState(42)
inter = Interactions(rs.randint(0, 943, 100000).astype(np.int32), rs.randint(0, 1682, 100000).astype(np.int32),
                     num_users=943, num_items=1682)
train, test = random_train_test_split(inter, random_state=np.random.RandomState(42))
from spotlight_amd.factorization import implicit as host  # noqa: E402

out = {'workload': 'C1 shape: 943 x 1682, 80000 train interactions, dim 32, bpr, batch 1024, 10 epochs (the reference\'s '
                   'tests/factorization/test_implicit.py:40-57 kwargs)', 'runs': []}
for name, kw in (('default Adam + l2 (dense: every row every step; per-minibatch launches)', dict(l2=1e-6)),
                 ('optimizer_func=Adagrad (row-sparse; persistent epoch kernel)',
                  dict(optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2)))):
    for pipelined in (True, False):
        host._PIPELINE_MAX_DRAWS = (1 << 22) if pipelined else 0
        mk = lambda: ImplicitFactorizationModel(loss='bpr', embedding_dim=32, batch_size=1024, n_iter=10, learning_rate=1e-2,
                                                use_cuda=True, random_state=np.random.RandomState(42), **kw)
        mk().fit(train)  # warm-up: library load, scratch
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            model = mk()
            t0 = time.perf_counter()
            model.fit(train)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        t1 = time.perf_counter()
        mrr = mrr_score(model, test, train=train).mean()
        t2 = time.perf_counter()
        out['runs'].append({'optimizer': name, 'next_epoch_prepared_while_training': pipelined, 'fit_s': best,
                            'interactions_per_s': len(train) * 10 / best,
                            'us_per_minibatch_end_to_end': best / (10 * ((len(train) + 1023) // 1024)) * 1e6,
                            'mrr_eval_s': t2 - t1, 'mrr_on_uniform_synthetic_data': float(mrr)})
print(json.dumps(out))
