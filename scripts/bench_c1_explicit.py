#!/usr/bin/env python
"""The reference README's quick-start at the MovieLens-100K SHAPE (the dataset itself cannot be downloaded here):
ExplicitFactorizationModel(n_iter=...) with its defaults -- regression loss, embedding_dim 32, batch_size 256, dense Adam
(spotlight/factorization/explicit.py:71-103) -- and with a row-sparse Adagrad, 943 users x 1682 items, 80 000 training
ratings, 10 epochs.  End-to-end fit() time with the persistent epoch kernel on / off and the epoch pipelining on / off."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.evaluation import rmse_score  # noqa: E402
from spotlight_amd.factorization import implicit as host  # noqa: E402
from spotlight_amd.factorization.explicit import ExplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402

rs = np.random.RandomState(42)
n = 80000
train = Interactions(rs.randint(0, 943, n).astype(np.int32), rs.randint(0, 1682, n).astype(np.int32),
                     ratings=rs.randint(1, 6, n).astype(np.float32), num_users=943, num_items=1682)
out = {'workload': 'explicit feedback, MovieLens-100K shape: 943 x 1682, 80000 ratings, dim 32, regression, batch 256, 10 epochs',
       'runs': []}
for name, kw in (('default Adam (dense: every row every step)', dict()),
                 ('optimizer_func=Adagrad (row-sparse)', dict(optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2)))):
    for epoch_kernel, pipelined in ((1, True), (1, False), (0, False)):
        os.environ['SPOTLIGHT_HIP_OPTIONS'] = 'epoch_kernel=%d' % epoch_kernel
        host._ENGINES.clear()
        host._PREP.clear()
        host._PIPELINE_MAX_DRAWS = (1 << 22) if pipelined else 0
        mk = lambda: ExplicitFactorizationModel(loss='regression', embedding_dim=32, batch_size=256, n_iter=10, use_cuda=True,
                                                random_state=np.random.RandomState(42), **kw)
        mk().fit(train)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            model = mk()
            t0 = time.perf_counter()
            model.fit(train)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out['runs'].append({'optimizer': name, 'persistent_epoch_kernel': bool(epoch_kernel),
                            'next_epoch_prepared_while_training': pipelined, 'fit_s': best,
                            'interactions_per_s': n * 10 / best, 'us_per_minibatch_end_to_end': best / (10 * ((n + 255) // 256)) * 1e6,
                            'train_rmse_on_uniform_synthetic_data': float(rmse_score(model, train))})
print(json.dumps(out))
