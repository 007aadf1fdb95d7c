"""--workload c1: BASELINE.json configs[0] -- the reference's own operating point, as its test suite runs it
(tests/factorization/test_implicit.py:40-57 of the reference): ImplicitFactorizationModel(loss='bpr', embedding_dim=32,
batch_size=1024, n_iter=10, learning_rate=1e-2, l2=1e-6) with the DEFAULT optimizer (dense Adam + l2: every row of every table
moves every step), on the MovieLens-100K SHAPE (943 users x 1682 items, 100 000 interactions, 80 / 20 split; the dataset itself
cannot be downloaded here: synthetic uniform ids from RandomState(42)).  The unit is the metric's own definition -- SURVEY.md 8(d):
len(interactions) x n_iter / wall(fit) -- of the WHOLE drop-in fit() call: id upload, ten numpy-exact shuffles, 790 minibatches
inside the persistent kernel, ten loss read-backs.  The reference's CPU fit() of the same call is timed by the cpu_baseline leg
(benchlib/cpu.py) and copied beside this record by bench.py."""
import time

import numpy as np
import torch

from benchlib.common import HBM_PEAK_GBS


def c1_interactions():
    from spotlight_amd.cross_validation import random_train_test_split
    from spotlight_amd.interactions import Interactions
    rs = np.random.RandomState(42)
    inter = Interactions(rs.randint(0, 943, 100000).astype(np.int32), rs.randint(0, 1682, 100000).astype(np.int32),
                         num_users=943, num_items=1682)
    return random_train_test_split(inter, random_state=np.random.RandomState(42))


def bench_c1(args):
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    torch.cuda.set_device(0)
    train, _ = c1_interactions()
    U, I, D, B, E = 943, 1682, 32, 1024, 10
    variants = {'default_dense_adam': dict(l2=1e-6),
                'sparse_adagrad': dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2))}
    n_mb = E * ((len(train) + B - 1) // B)
    out_v = {}
    for name, kw in variants.items():
        mk = lambda: ImplicitFactorizationModel(loss='bpr', embedding_dim=D, batch_size=B, n_iter=E, learning_rate=1e-2,
                                                use_cuda=True, random_state=np.random.RandomState(42), **kw)
        mk().fit(train)  # warm-up fit: library load, scratch, the prep lane's ctx
        torch.cuda.synchronize()
        times = []
        for _ in range(max(1, min(args.steps, 5))):
            model = mk()
            t0 = time.perf_counter()
            model.fit(train)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        best = min(times)
        # algorithmic bytes per minibatch: the row accesses of SURVEY.md 8(d) with S optimizer-state words (48 D + 64 for Adagrad,
        # 72 D + 88 with Adam's two moments) x B, plus -- dense Adam only -- the full-table update: every parameter and both its
        # moments read and written, its gradient read (28 B per parameter per step)
        params = (U + I) * (D + 1)
        if name == 'default_dense_adam':
            alg_mb = B * (72 * D + 88) + params * 28
        else:
            alg_mb = B * (48 * D + 64)
        out_v[name] = {'fit_seconds': best, 'fit_seconds_all': times, 'interactions_per_s': len(train) * E / best,
                       'us_per_minibatch_end_to_end': best / n_mb * 1e6, 'alg_bytes_per_minibatch': alg_mb,
                       'frac': alg_mb * n_mb / best / 1e9 / HBM_PEAK_GBS}
    head = out_v['default_dense_adam']
    return {'metric': 'training interactions/sec, whole fit(), MovieLens-100K shape, BPR dim=32', 'value': head['interactions_per_s'],
            'unit': 'interactions/s', 'n_gpus': 1, 'steps': len(head['fit_seconds_all']), 'warmup': 1,
            'ms_per_step': head['fit_seconds'] / n_mb * 1e3, 'higher_is_better': True, 'dtype': 'f32',
            'data': 'synthetic (MovieLens-100K shape; the dataset cannot be downloaded here)',
            'config': {'workload': 'C1: ImplicitFactorizationModel(loss=bpr, embedding_dim=32, batch_size=1024, n_iter=10, '
                                   'learning_rate=1e-2, l2=1e-6).fit() on 943 users x 1682 items, %d train interactions '
                                   '(the reference\'s tests/factorization/test_implicit.py:40-57); a step = one minibatch of the '
                                   'whole call: min of %d timed fit()s after a warm-up fit()' % (len(train), len(head['fit_seconds_all']))},
            'roofline': {'bound': 'launch / barrier latency (87 K parameters: two grid barriers per minibatch inside the persistent kernel)',
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'alg_bytes_per_interaction': head['alg_bytes_per_minibatch'] / B,
                         'step_frac_of_peak': head['frac'], 'variants': out_v}}
