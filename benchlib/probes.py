"""Measured bandwidth probes, the world-1 consistency run of the row-sharded path, and the end-to-end fit() leg of the
headline workload."""
import time

import numpy as np
import torch

from spotlight_amd import _native


def measured_stream_rates(be, stream):
    """Copy / triad GB/s of this GPU (slk_probe_stream: float4 kernels over 1 GiB buffers) -- the measured figure
    SURVEY.md 8(d) asks for next to the nominal peak."""
    n = (1 << 28) if be.kind == 'hip' else (1 << 12)
    a, b, c = (torch.ones(n, device=be.dev) for _ in range(3))
    ms = {k: be.engine.probe_stream(k, a.data_ptr(), b.data_ptr(), c.data_ptr(), n, iters=10, stream=stream) for k in range(12)}
    del a, b, c
    gbs = {'copy_plain': 8.0 * n / ms[0] / 1e6, 'triad_plain': 12.0 * n / ms[1] / 1e6, 'copy_nt_x4': 8.0 * n / ms[2] / 1e6,
           'triad_nt_x4': 12.0 * n / ms[3] / 1e6, 'read_only': 4.0 * n / ms[4] / 1e6, 'write_only': 4.0 * n / ms[5] / 1e6,
           'copy_chunk_x8': 8.0 * n / ms[6] / 1e6, 'copy_chunk_x8_nt': 8.0 * n / ms[7] / 1e6, 'copy_chunk_x4_16wg': 8.0 * n / ms[8] / 1e6,
           'copy_chunk_x16_4wg': 8.0 * n / ms[9] / 1e6, 'copy_chunk_x8_ntload': 8.0 * n / ms[10] / 1e6,
           'copy_chunk_x4_nt_32wg': 8.0 * n / ms[11] / 1e6}
    return {'copy_GBs': max(v for k, v in gbs.items() if k.startswith('copy')),
            'triad_GBs': max(gbs['triad_plain'], gbs['triad_nt_x4']), 'variants_GBs': gbs,
            'note': 'slk_probe_stream over 1 GiB buffers, hipEvents, 10 launches each: float4 copy / triad, plain grid-stride and '
                    'non-temporal with 4 accesses in flight per lane; read-only and write-only streams; chunked copies (a workgroup '
                    'moves contiguous 16-64 KB chunks, 4-16 loads in flight per lane, plain / non-temporal); copy_GBs = the best copy'}


def sharded_world1_check(be, args, tables, s1, s2, users, items, B, stream):
    """N = 1 consistency of the two engines: the same two minibatches, from the same tables and the same RNG
    state, through the fused path and through the row-sharded exchange path at world 1 (exchange = device copy);
    per-minibatch losses must agree.  Runs on copies of the tables."""
    from spotlight_amd.factorization.sharded import ShardedBilinearTrainer
    import torch.distributed as dist
    eng = be.engine
    K = 2
    state = np.random.RandomState(77).get_state()
    losses = []
    times = []
    for path in ('fused', 'sharded'):
        t = [x.clone() for x in tables]
        a1 = [x.clone() for x in s1]
        a2 = [x.clone() for x in s2] if s2 else None
        op = _native.make_optim(args.opt, [x.data_ptr() for x in a1], [x.data_ptr() for x in a2] if a2 else None, lr=1e-2,
                                weight_decay=1e-6 if args.opt == 'adam_dense' else 0.0)
        mb = torch.zeros(K, device=be.dev)
        eng.rng_set_state(state)
        if path == 'fused':
            tb = _native.make_tables([x.data_ptr() for x in t], t[0].shape[0], t[1].shape[0], args.dim)
            run = lambda lo: eng.bilinear_train(tb, op, users[lo:].data_ptr(), items[lo:].data_ptr(), K * B, B, args.loss, 1,
                                                mb.data_ptr(), stream=stream)
        else:
            tr = ShardedBilinearTrainer(eng, t, op, t[1].shape[0], stream=stream, slices=args.slices or None)
            tr.reserve(B, K)
            run = lambda lo: tr.train(users[lo:lo + K * B], items[lo:lo + K * B], B, loss=args.loss, mb_loss=mb)
        run(0)
        be.sync()
        first = mb.cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        run(K * B)  # the same call again on the next minibatches: buffers allocated, code paths warm
        be.sync()
        times.append((time.perf_counter() - t0) / K * 1e3)
        mb.copy_(torch.from_numpy(first).to(mb.dtype))
        losses.append(mb.cpu().numpy().astype(np.float64))
        del t, a1, a2
    rel = float(np.abs(losses[0] - losses[1]).max() / np.abs(losses[0]).max())
    return {'minibatches': K, 'loss_fused': losses[0].tolist(), 'loss_sharded_world1': losses[1].tolist(),
            'max_rel_diff': rel, 'consistent': bool(rel <= 1e-5),
            'ms_per_step_second_call': {'fused': times[0], 'sharded_world1': times[1]}}


def fit_end_to_end(be, args):
    """The drop-in API around the engine, end to end: ImplicitFactorizationModel.fit() (spotlight/factorization/implicit.py:184-252)
    on the workload's shapes -- per epoch the numpy-exact device shuffle, the id gathers, every minibatch, the loss read-back; the
    ids are uploaded once per fit() (host -> HBM, included).  One warm fit() of 3 epochs first (table initialisation, scratch,
    the epoch loop's id buffers), then a timed fit() of 10 epochs (the reference's default n_iter)."""
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    n = int(args.fit_interactions)
    rs = np.random.RandomState(5)
    inter = Interactions(rs.randint(0, args.users, n).astype(np.int32), rs.randint(0, args.items, n).astype(np.int32),
                         num_users=args.users, num_items=args.items)
    opts = {'adagrad': dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2)),
            'sparse_adam': dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=1e-2)),
            'adam_dense': dict(l2=1e-6)}[args.opt]
    # (the warm fit runs 3 epochs: the large-epoch loop rotates three pairs of id buffers, and the timed fit should find all of
    # them in torch's caching allocator like every fit() after a process's first -- fresh HIP allocations of that size cost
    # 15-25 ms each, profiles/r04_t_fit_first_epoch_probe.txt)
    model = ImplicitFactorizationModel(loss=args.loss, embedding_dim=args.dim, n_iter=3, batch_size=args.batch, use_cuda=True,
                                       random_state=np.random.RandomState(1), **opts)
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    first = time.perf_counter() - t0
    epochs = 10  # the reference's default n_iter: the id upload and the first epoch's unhidden shuffle amortise as they do for a user
    model._n_iter = epochs
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    t_full = time.perf_counter() - t0
    dt = t_full / epochs
    # the same call with 2 epochs: the difference is 8 epochs of the steady state (no id upload, no first shuffle, no drain)
    model._n_iter = 2
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    t_two = time.perf_counter() - t0
    steady = (t_full - t_two) / (epochs - 2)
    return {'interactions_per_epoch': n, 'epochs_timed': epochs, 'seconds_per_epoch': dt, 'interactions_per_s': n / dt,
            'steady_state_seconds_per_epoch': steady, 'steady_state_interactions_per_s': n / steady,
            'steady_state_note': '(fit of 10 epochs - fit of 2 epochs) / 8: what every further epoch costs once the three-stage '
                                 'pipeline runs (next epoch\'s negatives + first sorts, the shuffle after next, this epoch\'s passes)',
            'first_fit_seconds': first,
            'what': 'ImplicitFactorizationModel.fit(): id upload (once per fit), per epoch the numpy-exact device shuffle + id '
                    'gathers + %d minibatches + the loss read-back; first_fit_seconds also holds table initialisation on the '
                    'host and scratch allocation' % ((n + args.batch - 1) // args.batch)}
