"""bench.py's cpu_baseline leg: the reference itself on the box's host cores (oracle/_ref, kind "reference") and the scalar
C port (oracle/slk_oracle.c, kind "port").  The only place outside tests/ and smoke() that touches oracle/."""
import json
import os
import sys
import time

import numpy as np

from benchlib.common import ROOT


def reference_cpu_baseline(args, seconds):
    """Spotlight's own CPU PyTorch path (the copy staged by oracle/make_ref.sh under oracle/_ref/) timed
    on this machine's host cores by oracle/ref_cpu_baseline.py, in its own process: same table shapes, loss
    and minibatch as the GPU workload, protocol of the reference's examples/bloom_embeddings/performance.py:24-38.
    Returns None when the copy is not staged."""
    import subprocess
    script = os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py')
    if not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'spotlight')):
        return None
    cmd = [sys.executable, script, '--users', str(args.users), '--items', str(args.items), '--dim', str(args.dim),
           '--batch', str(args.batch), '--loss', args.loss, '--seconds', str(seconds)]
    default_c2 = (args.users, args.items, args.dim, args.batch, args.loss, args.opt) == (10_000_000, 1_000_000, 64, 1 << 20, 'bpr', 'adagrad')
    if default_c2 and not getattr(args, 'no_configs', False):
        cmd += ['--c1', '1', '--hip', '1']  # the reference's own operating point on the CPU; the reference on the HIP device (stock PyTorch-ROCm)
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150 + 12 * seconds)
        rec = json.loads(res.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
        return {'error': repr(e)[:300]}
    if 'sparse_adagrad' not in rec:
        return {'error': str(rec)[:300]}
    sa, da = rec['sparse_adagrad'], rec.get('default_dense_adam')
    out = {'value': sa['interactions_per_s'], 'unit': 'interactions/s', 'cores': sa['threads'], 'kind': 'reference',
           'cpu_model': rec['cpu_model'], 'host_cores': rec['host_cores'],
           'interactions_per_s_by_threads': sa['interactions_per_s_by_threads'],
           'sample': 'spotlight ImplicitFactorizationModel.fit() on CPU PyTorch %s, sparse=True + Adagrad(lr=1e-2), %s loss, '
                     '%d users x %d items, dim %d, minibatch %d (the GPU workload uses %d): warm-up fit + min of 2 timed fits of %d '
                     'minibatch(es) (%.1f s each); torch.set_num_threads: every host core (%d, on 32768 interactions) and 16 were '
                     'probed, the faster (%d) was timed%s'
                     % (rec['torch'], rec['loss'], rec['users'], rec['items'], rec['dim'], sa.get('batch', rec['batch']), rec['gpu_workload_batch'],
                        sa['minibatches_per_fit'], sa['seconds'], rec['host_cores'], sa['threads'],
                        '; ' + rec['note'] if rec['note'] else '')}
    if 'c1_fit' in rec:
        out['c1_reference_fit'] = rec['c1_fit']
    if 'hip_sparse_adagrad' in rec:
        out['reference_on_hip'] = rec['hip_sparse_adagrad']
    if da:
        out['reference_default_dense_adam'] = {'value': da['interactions_per_s'], 'unit': 'interactions/s', 'cores': da['threads'],
                                               'sample': '%d minibatch(es) of %d per fit, %.1f s' % (da['minibatches_per_fit'], da.get('batch', rec['batch']), da['seconds'])}
    return out


def cpu_baseline(args, seconds):
    """The oracle (CPU port of spotlight/factorization/implicit.py:223-243 with a row-sparse
    Adagrad, i.e. the reference's sparse=True + Adagrad path) on the same table shapes."""
    from oracle.oracle import BilinearOracle, Rng
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 16 << 30
    U, I, D = args.users, args.items, args.dim
    need = (U + I) * D * 4 * 5
    note = ''
    while need > 0.6 * avail and U > 100_000:
        U //= 2
        need = (U + I) * D * 4 * 5
        note = ' (user table scaled to %d rows to fit host RAM)' % U
    rs = np.random.default_rng(0)
    block = (rs.standard_normal(1 << 20, dtype=np.float32) / D)
    p = [np.resize(block, (U, D)), np.resize(block, (I, D)), np.zeros(U, np.float32), np.zeros(I, np.float32)]
    ora = BilinearOracle(*p, opt=args.opt, lr=1e-2, sparse_grads=True)
    del p
    rng = Rng(seed=1)
    B = min(args.batch, 1 << 18)
    done, t_total = 0, 0.0
    # warm the page tables of the gradient buffers with one untimed minibatch
    users, items = rs.integers(0, U, B), rs.integers(0, I, B)
    ora.train(rng, users, items, B, loss=args.loss)
    while t_total < seconds and done < (1 << 24):
        users, items = rs.integers(0, U, B), rs.integers(0, I, B)
        t0 = time.perf_counter()
        ora.train(rng, users, items, B, loss=args.loss)
        t_total += time.perf_counter() - t0
        done += B
    return {'value': done / t_total, 'unit': 'interactions/s', 'cores': 1, 'kind': 'port',
            'sample': '%d minibatches of %d interactions, same table shapes%s, oracle/slk_oracle.c '
                      'single thread (%d host cores present)' % (done // B, B, note, os.cpu_count())}
