#!/usr/bin/env python
"""MEASUREMENT INFRASTRUCTURE (bench.py's `cpu_baseline` leg) -- not part of the product.

Times the reference ITSELF -- spotlight.factorization.implicit.ImplicitFactorizationModel.fit() on CPU
PyTorch, imported from the copy staged by oracle/make_ref.sh under oracle/_ref/ -- on the host cores of
the machine bench.py runs on, on the same table shapes / loss / minibatch as the GPU workload:

  protocol    examples/bloom_embeddings/performance.py:24-38 of the reference: one warm-up fit(), then the
              minimum wall time of 2 timed fit()s of the same model on the same Interactions
  threads     torch.set_num_threads(os.cpu_count()), stated in the output
  variants    sparse_adagrad      sparse=True + optimizer_func=Adagrad(lr=1e-2): the algorithmic equivalent
                                  of the GPU step (row-sparse gradients, row-sparse update)
              default_dense_adam  the reference's defaults (dense gradients, Adam over every table row
                                  every minibatch; factorization/implicit.py:144-148)
  sample      bounded: the number of interactions per fit() is chosen from the warm-up's rate so that the
              whole leg takes about --seconds

Prints one JSON object.  Runs in its own process (bench.py spawns it) so that its 2 x (tables + optimizer
state + gradients) of host memory and its thread pool are gone before anything else happens.
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, '_ref')


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=10_000_000)
    ap.add_argument('--items', type=int, default=1_000_000)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--batch', type=int, default=1 << 20)
    ap.add_argument('--loss', default='bpr')
    ap.add_argument('--seconds', type=float, default=24.0)
    ap.add_argument('--threads', type=int, default=0, help='0: os.cpu_count()')
    ap.add_argument('--variants', default='sparse_adagrad,default_dense_adam')
    args = ap.parse_args()

    if not os.path.isdir(os.path.join(REF, 'spotlight')):
        print(json.dumps({'error': 'oracle/_ref/spotlight is not staged: run `sh oracle/make_ref.sh` where '
                                   '/root/reference exists (build() does)'}))
        return 2
    sys.path.insert(0, REF)
    import numpy as np
    import torch
    threads = args.threads or os.cpu_count()
    torch.set_num_threads(threads)
    from spotlight.factorization.implicit import ImplicitFactorizationModel
    from spotlight.interactions import Interactions

    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 32 << 30
    U, I, D, B = args.users, args.items, args.dim, args.batch
    note = ''
    # dense Adam holds param + grad + 2 moments (+ temporaries) of every table
    while (U + I) * D * 4 * 7 > 0.7 * avail and U > 100_000:
        U //= 2
        note = ' (user table scaled to %d rows to fit host RAM)' % U

    rs = np.random.RandomState(0)
    n_max = 16 * B
    users = rs.randint(0, U, n_max).astype(np.int32)
    items = rs.randint(0, I, n_max).astype(np.int32)

    def inter(n):
        return Interactions(users[:n], items[:n], num_users=U, num_items=I)

    out = {'threads': threads, 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'users': U, 'items': I, 'dim': D,
           'batch': B, 'loss': args.loss, 'note': note.strip(), 'torch': torch.__version__,
           'protocol': 'warm-up fit() + min of 2 timed fit()s (reference examples/bloom_embeddings/performance.py:24-38)'}
    variants = [v for v in args.variants.split(',') if v]
    share = args.seconds / max(len(variants), 1)
    for name in variants:
        kw = (dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2))
              if name == 'sparse_adagrad' else dict())
        model = ImplicitFactorizationModel(loss=args.loss, embedding_dim=D, n_iter=1, batch_size=B,
                                           random_state=np.random.RandomState(1), **kw)
        t0 = time.perf_counter()
        model.fit(inter(B))  # warm-up epoch: table initialisation, allocator, thread pool
        t_warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        model.fit(inter(B))  # rate probe (initialisation excluded)
        per_mb = time.perf_counter() - t0
        # probe + 2 timed fits ~ this variant's share of --seconds (the warm-up's one-off costs come on top)
        k = int(max(1, min(n_max // B, share / 3.0 / max(per_mb, 1e-6))))
        data = inter(k * B)
        timings = []
        for _ in range(2):
            t0 = time.perf_counter()
            model.fit(data)
            timings.append(time.perf_counter() - t0)
        out[name] = {'interactions_per_fit': k * B, 'minibatches_per_fit': k, 'seconds': min(timings),
                     'timings': timings, 'warmup_seconds': t_warm, 'interactions_per_s': k * B / min(timings)}
        del model
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
