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
mk = lambda: ImplicitFactorizationModel(loss='bpr', embedding_dim=32, batch_size=1024, n_iter=10, learning_rate=1e-2,
                                        l2=1e-6, use_cuda=True, random_state=np.random.RandomState(42))
mk().fit(train)  # warm-up: library load, scratch
torch.cuda.synchronize()
model = mk()
t0 = time.perf_counter()
model.fit(train)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
mrr = mrr_score(model, test, train=train).mean()
t2 = time.perf_counter()
print(json.dumps({'workload': 'C1 shape: 943 x 1682, 80000 train interactions, dim 32, bpr, default Adam + l2, batch 1024, '
                              '10 epochs', 'fit_s': dt, 'interactions_per_s': len(train) * 10 / dt,
                  'us_per_minibatch': dt / (10 * ((len(train) + 1023) // 1024)) * 1e6, 'mrr_eval_s': t2 - t1,
                  'mrr_on_uniform_synthetic_data': float(mrr)}))
