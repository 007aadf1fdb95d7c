#!/usr/bin/env python
"""Same-process A/B of the drop-in fit() under different engine options (C2 shapes, 2^25 interactions x 10 epochs):
one model, one warm fit(), then every configuration timed `--repeat` times, the first one again at the end.

    python scripts/ab_fit_options.py --configs prefetch_wait=1 prefetch_wait=0 fit:overlap_prep=2

`name=value` sets a ctx option for the configuration; `fit:name=value` changes what fit() itself asks of the ctx for its
duration (spotlight_amd/factorization/implicit.py::_FIT_OPTIONS)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization import implicit as host  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1 << 25)
    ap.add_argument('--epochs', type=int, default=10)
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--configs', nargs='*', default=[])
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    U, I, n = 10_000_000, 1_000_000, args.n
    rs = np.random.RandomState(5)
    inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=3, batch_size=1 << 20, use_cuda=True, sparse=True,
                                       optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2), random_state=np.random.RandomState(1))
    model.fit(inter)
    torch.cuda.synchronize()
    eng = host._engine_for(torch.device('cuda', torch.cuda.current_device()))
    base_fit = dict(host._FIT_OPTIONS)
    real_shuffle = host.device_epoch_shuffle
    out_f = open(args.out, 'a') if args.out else None
    configs = list(args.configs)
    configs = configs + configs[:1]
    for label in configs:
        ctx_opts, fit_opts = {}, dict(base_fit)
        dbg_shuffle = 0
        for kv in filter(None, label.split(',')):
            k, v = kv.rsplit("=", 1)
            if k.startswith('fit:') or k.startswith('fit.'):  # ('fit.': scripts/gpu_run.sh splits its stage arguments at ':')
                fit_opts[k[4:]] = int(v)
            elif k == 'dbg.shuffle':
                # MEASUREMENT ONLY (results are not the reference's): 1 = the epoch permutation is the identity, written by one
                # streaming kernel -- what an epoch costs WITHOUT the numpy-exact shuffle's work beside its passes (the id gathers
                # still run); 0 = the real shuffle
                dbg_shuffle = int(v)
            else:
                ctx_opts[k] = int(v)
        host._FIT_OPTIONS.clear()
        host._FIT_OPTIONS.update(fit_opts)
        if dbg_shuffle:
            def identity_shuffle(engine, random_state, n_, d_perm, arrays, stream):
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    torch.arange(n_, out=d_perm)
                real_perm = engine.shuffle_perm
                engine.shuffle_perm = lambda *a, **k: None
                try:
                    real_shuffle(engine, random_state, n_, d_perm, arrays, stream)
                finally:
                    engine.shuffle_perm = real_perm
            host.device_epoch_shuffle = identity_shuffle
        else:
            host.device_epoch_shuffle = real_shuffle
        with eng.options(**ctx_opts):
            times, steady = [], []
            for _ in range(args.repeat):
                model._n_iter = args.epochs
                t0 = time.perf_counter()
                model.fit(inter)
                torch.cuda.synchronize()
                t_full = time.perf_counter() - t0
                model._n_iter = 2
                t0 = time.perf_counter()
                model.fit(inter)
                torch.cuda.synchronize()
                t_two = time.perf_counter() - t0
                times.append(t_full / args.epochs)
                steady.append((t_full - t_two) / (args.epochs - 2))
        rec = {'label': label, 'ms_per_epoch': [t * 1e3 for t in times], 'steady_ms_per_epoch': [t * 1e3 for t in steady],
               'G_interactions_per_s': n / min(times) / 1e9, 'steady_G_interactions_per_s': n / min(steady) / 1e9}
        line = json.dumps(rec)
        print(line, flush=True)
        if out_f:
            out_f.write(line + '\n')
            out_f.flush()
    host._FIT_OPTIONS.clear()
    host._FIT_OPTIONS.update(base_fit)
    host.device_epoch_shuffle = real_shuffle


if __name__ == '__main__':
    main()
