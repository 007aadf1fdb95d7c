"""TEST HARNESS: one rank of the drop-in sharded model check (tests/test_sharded.py).

Every rank fits spotlight_amd.factorization.sharded.ShardedImplicitFactorizationModel on the same
Interactions with the same seed; rank 0 also fits the single-device ImplicitFactorizationModel
and compares: identical RandomState afterwards (same shuffles, same negatives), tables equal to
summation-order noise, predictions equal.  Backend 'emu' (gloo + emulator) or 'hip' (nccl)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from spotlight_amd import _native  # noqa: E402
from spotlight_amd.factorization import implicit as host  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.factorization.sharded import ShardedImplicitFactorizationModel, local_rows  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    backend, loss, opt = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if backend == 'emu':
        from emu_backend import emu_lib
        dist.init_process_group('gloo')
        eng = _native.Engine(0, lib=emu_lib())
        host._engine_for = lambda device: eng
        host._stream_for = lambda device: 0
        host._model_device = lambda: torch.device('cpu')
    else:
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', rank))

    U, I, N, D, B = 61, 47, 400, 16, 96
    rs = np.random.RandomState(3)
    inter = Interactions(rs.randint(0, U, N).astype(np.int32), rs.randint(0, I, N).astype(np.int32),
                         num_users=U, num_items=I)
    of = {'adagrad': lambda p: torch.optim.Adagrad(p, lr=0.05, initial_accumulator_value=0.1),
          'adam': None,
          'sparse_adam': lambda p: torch.optim.SparseAdam(list(p), lr=0.01)}[opt]
    kw = dict(loss=loss, embedding_dim=D, n_iter=2, batch_size=B, l2=1e-6, optimizer_func=of,
              sparse=(opt == 'sparse_adam'))
    model = ShardedImplicitFactorizationModel(random_state=np.random.RandomState(42), **kw)
    floor = os.environ.get('SHARD_TEST_SHADOW_FLOOR')  # local item shards of at least this many rows train bias-shadowed
    if floor:
        host._BIAS_SHADOW_MIN_ITEMS = int(floor)
    calls0 = host._engine_for(torch.device('cpu') if backend == 'emu' else torch.device('cuda', rank)).get_stat('shadowed_calls')
    model.fit(inter)
    if backend == 'hip':
        # resume under ANOTHER current stream (the cached trainer kept the first fit()'s stream and refused): the
        # trainer follows the stream in use, the collectives and the kernels stay ordered on it
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model.fit(inter)
        torch.cuda.current_stream().wait_stream(side)
    else:
        model.fit(inter)  # resume
    if floor:
        eng_ = host._engine_for(torch.device('cpu') if backend == 'emu' else torch.device('cuda', rank))
        n_mb = (N + B - 1) // B
        assert eng_.get_stat('shadowed_calls') - calls0 == (2 * 2 * n_mb if opt == 'adagrad' else 0)  # every item pass of both fits
        host._BIAS_SHADOW_MIN_ITEMS = 1 << 40  # (the one-GPU model below trains plain)
    pred_all = model.predict(5)
    pu, pi = np.arange(0, 20, dtype=np.int64), (np.arange(0, 20, dtype=np.int64) * 7 + 1) % I
    pred_pairs = model.predict(pu, pi)
    assert model._net.tables()[0].shape[0] == local_rows(U, world, rank)
    # ranking metrics on the sharded model go through predict() (every rank makes the same calls)
    from spotlight_amd.evaluation import mrr_score
    test_inter = Interactions(rs.randint(0, 9, 30).astype(np.int32), rs.randint(0, I, 30).astype(np.int32),
                              num_users=U, num_items=I)
    mrr = mrr_score(model, test_inter, train=inter)

    # reassemble the tables
    full = []
    for t, loc in enumerate(model._net.tables()):
        rows_total = U if t in (0, 2) else I
        pad = max(local_rows(rows_total, world, r) for r in range(world))
        buf = torch.zeros((pad,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
        buf[:loc.shape[0]] = loc.detach()
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        whole = np.zeros((rows_total,) + tuple(loc.shape[1:]), dtype=np.float32)
        for r in range(world):
            whole[r::world] = gathered[r][:local_rows(rows_total, world, r)].cpu().numpy()
        full.append(whole)

    if rank == 0:
        ref = ImplicitFactorizationModel(random_state=np.random.RandomState(42), **kw)
        ref.fit(inter)
        ref.fit(inter)
        a, b = model._random_state.get_state(), ref._random_state.get_state()
        assert (a[1] == b[1]).all() and a[2] == b[2], 'RandomState consumption differs'
        for t, w in enumerate(ref._net.tables()):
            want = w.detach().cpu().numpy()
            scale = max(np.abs(want).max(), 1e-3)
            if opt == 'adagrad':
                err = np.abs(full[t] - want).max() / scale
                assert err < 2e-4, (t, err)  # 20 optimizer steps; only the summation order of item rows differs
            else:
                # Adam normalises by sqrt(v): an element whose gradient is +-1/B cancellation noise moves
                # by O(lr) in a direction set by summation order -- judge the
                # trajectory by the fraction of elements outside tolerance, as engine_checks does
                bad = np.abs(full[t] - want) > 1e-3 * scale
                assert bad.mean() <= 0.05, (t, bad.mean())
        want_all, want_pairs = ref.predict(5), ref.predict(pu, pi)
        ptol = 1e-4 if opt == 'adagrad' else 5e-3
        assert np.abs(pred_all - want_all).max() <= ptol * np.abs(want_all).max()
        assert np.abs(pred_pairs - want_pairs).max() <= ptol * np.abs(want_pairs).max()
        assert pred_all.dtype == np.float32 and pred_all.shape == (I,)
        want_mrr = mrr_score(ref, test_inter, train=inter)  # the one-device model's fast path
        # scores agree to ptol, so nearly every rank (hence reciprocal rank) is identical
        assert mrr.shape == want_mrr.shape and np.isfinite(mrr).all()
        assert np.mean(np.abs(mrr - want_mrr) > 1e-6) <= (0.25 if opt == 'adagrad' else 0.5), (mrr, want_mrr)
        print('SHARD_MODEL_OK world=%d loss=%s opt=%s' % (world, loss, opt))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
