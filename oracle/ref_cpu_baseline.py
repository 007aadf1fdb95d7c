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
  sample      bounded: the number of interactions per fit() is chosen from the probed rate so that the timed
              fits take about --seconds in all (model construction, warm-up and thread probes come on top)

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
    ap.add_argument('--c1', type=int, default=0, help='1: also time the reference\'s CPU fit() of BASELINE.json configs[0] (MovieLens-100K shape)')
    ap.add_argument('--hip', type=int, default=0, help='1: also time the reference on the HIP device through stock PyTorch-ROCm (use_cuda=True)')
    ap.add_argument('--hip-minibatches', type=int, default=4)
    args = ap.parse_args()

    if not os.path.isdir(os.path.join(REF, 'spotlight')):
        print(json.dumps({'error': 'oracle/_ref/spotlight is not staged: run `sh oracle/make_ref.sh` where '
                                   '/root/reference exists (build() does)'}))
        return 2
    # `spotlight` = the staged reference, loaded from its explicit location (never this repository's alias package)
    os.environ.setdefault('SPOTLIGHT_REFERENCE', REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.reference_import import import_reference
    import_reference()
    import numpy as np
    import torch
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

    # The sparse-Adagrad variant -- the one the line's cpu_baseline.value is -- runs the GPU workload's OWN minibatch (2^20: 2-3 s
    # per minibatch at 16 threads); the secondary dense-Adam variant (9 s per 2^20 minibatch) keeps a bounded 2^18 one.  Both are
    # stated in the output.
    Bc = B
    rs = np.random.RandomState(0)
    n_max = 8 * Bc
    users = rs.randint(0, U, n_max).astype(np.int32)
    items = rs.randint(0, I, n_max).astype(np.int32)

    def inter(n):
        return Interactions(users[:n], items[:n], num_users=U, num_items=I)

    def timed_fit(model, data):
        t0 = time.perf_counter()
        model.fit(data)
        return time.perf_counter() - t0

    out = {'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'users': U, 'items': I, 'dim': D, 'batch': Bc,
           'gpu_workload_batch': B, 'loss': args.loss, 'note': note.strip(), 'torch': torch.__version__,
           'protocol': 'warm-up fit() + min of 2 timed fit()s (reference examples/bloom_embeddings/performance.py:24-38)'}
    variants = [v for v in args.variants.split(',') if v]
    all_threads = args.threads or os.cpu_count()
    B_full = Bc
    for vi, name in enumerate(variants):
        kw = (dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2))
              if name == 'sparse_adagrad' else dict())
        Bc = B_full if name == 'sparse_adagrad' else min(B_full, 1 << 18)
        few = min(all_threads, 16)
        torch.set_num_threads(few)
        model = ImplicitFactorizationModel(loss=args.loss, embedding_dim=D, n_iter=1, batch_size=Bc,
                                           random_state=np.random.RandomState(1), **kw)
        t_warm = timed_fit(model, inter(Bc))  # warm-up epoch: table initialisation, allocator, thread pool
        # the protocol's thread count is every host core; torch's sparse CPU ops do not scale to hundreds of threads, so a
        # moderate count is probed too and the better one is used for the timed fits (both rates are reported).  The
        # every-core probe runs an eighth of a minibatch: at 256 threads a whole one takes the leg's entire budget.
        rate = {few: Bc / timed_fit(model, inter(Bc))}
        if all_threads != few:
            torch.set_num_threads(all_threads)
            n_probe = max(min(Bc // 8, 1 << 15), 1)
            timed_fit(model, inter(n_probe))  # the larger thread pool's first use
            rate[all_threads] = n_probe / timed_fit(model, inter(n_probe))
        best = max(rate, key=rate.get)
        torch.set_num_threads(best)
        per_mb = Bc / rate[best]
        # --seconds is the budget of the TIMED fits (two per variant); model construction, warm-up and probes come on top
        k = int(max(1, min(n_max // Bc, args.seconds / len(variants) / 2.0 / max(per_mb, 1e-6))))
        data = inter(k * Bc)
        timings = [timed_fit(model, data) for _ in range(2)]
        out[name] = {'interactions_per_fit': k * Bc, 'minibatches_per_fit': k, 'batch': Bc, 'seconds': min(timings), 'timings': timings,
                     'warmup_seconds': t_warm, 'threads': best, 'interactions_per_s': k * Bc / min(timings),
                     'interactions_per_s_by_threads': {str(th): r for th, r in rate.items()}}
        del model
    out['threads'] = out[variants[0]]['threads'] if variants else all_threads

    if args.c1:
        # BASELINE.json configs[0], the reference's own test configuration (tests/factorization/test_implicit.py:40-57) on the
        # MovieLens-100K shape: the WHOLE fit() call, the reference's defaults (dense Adam + l2), on this machine's host cores --
        # what bench.py's `configs.c1` leg runs through the drop-in API on the GPU.  Same ids: RandomState(42), the reference's
        # own random_train_test_split.
        from spotlight.cross_validation import random_train_test_split
        rs1 = np.random.RandomState(42)
        full = Interactions(rs1.randint(0, 943, 100000).astype(np.int32), rs1.randint(0, 1682, 100000).astype(np.int32),
                            num_users=943, num_items=1682)
        train, _ = random_train_test_split(full, random_state=np.random.RandomState(42))
        c1 = {}
        mk = lambda: ImplicitFactorizationModel(loss='bpr', embedding_dim=32, batch_size=1024, n_iter=10, learning_rate=1e-2,
                                                l2=1e-6, random_state=np.random.RandomState(42))
        torch.set_num_threads(1)
        timed_fit(mk(), train)  # warm-up fit
        for th in sorted({1, min(all_threads, 8)}):  # (tables of 87 K parameters: more threads only add synchronisation)
            torch.set_num_threads(th)
            c1[th] = min(timed_fit(mk(), train) for _ in range(2))
        best = min(c1, key=c1.get)
        out['c1_fit'] = {'fit_seconds': c1[best], 'threads': best, 'fit_seconds_by_threads': {str(k): v for k, v in c1.items()},
                         'interactions_per_s': len(train) * 10 / c1[best], 'train_interactions': len(train),
                         'what': 'spotlight ImplicitFactorizationModel(loss=bpr, embedding_dim=32, batch_size=1024, n_iter=10, '
                                 'learning_rate=1e-2, l2=1e-6).fit() on CPU PyTorch, 943 x 1682, warm-up fit + min of 2'}

    if args.hip and torch.cuda.is_available():
        # the reference ITSELF on the HIP device through stock PyTorch-ROCm (use_cuda=True) -- sparse=True +
        # Adagrad, the algorithmic equivalent of the fused step -- on the GPU workload's shapes and minibatch, a bounded number
        # of minibatches.  A second baseline, not the target: host-side numpy shuffle + per-minibatch negative draw + H2D as
        # the reference does them, torch's gather / index_add / coalesce / sparse Adagrad kernels on the device.
        try:
            torch.set_num_threads(min(all_threads, 16))
            U2, I2 = args.users, args.items
            k = max(1, int(args.hip_minibatches))
            model = ImplicitFactorizationModel(loss=args.loss, embedding_dim=D, n_iter=1, batch_size=B_full, use_cuda=True, sparse=True,
                                               optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2),
                                               random_state=np.random.RandomState(1))
            rs2 = np.random.RandomState(0)
            data = Interactions(rs2.randint(0, U2, k * B_full).astype(np.int32), rs2.randint(0, I2, k * B_full).astype(np.int32),
                                num_users=U2, num_items=I2)
            t_warm = timed_fit(model, data)
            torch.cuda.synchronize()
            timings = []
            for _ in range(2):
                t0 = time.perf_counter()
                model.fit(data)
                torch.cuda.synchronize()
                timings.append(time.perf_counter() - t0)
            out['hip_sparse_adagrad'] = {'interactions_per_s': k * B_full / min(timings), 'ms_per_minibatch': min(timings) / k * 1e3,
                                         'minibatches_per_fit': k, 'batch': B_full, 'timings': timings, 'warmup_seconds': t_warm,
                                         'device': torch.cuda.get_device_name(0), 'users': U2, 'items': I2,
                                         'what': 'spotlight ImplicitFactorizationModel(use_cuda=True, sparse=True, optimizer_func=Adagrad'
                                                 '(lr=1e-2)).fit() through stock PyTorch-ROCm ops on the same GPU: warm-up fit + min of '
                                                 '2 timed fits (host shuffle, per-minibatch numpy negatives + H2D included, as the '
                                                 'reference runs)'}
        except Exception as e:  # noqa: BLE001 -- a reported extra
            out['hip_sparse_adagrad'] = {'error': repr(e)[:300]}
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
