"""TEST HARNESS: one rank of the row-sharded training parity check (tests/test_sharded.py).

Backend 'emu': the fiber-emulator build of the engine sources + gloo on CPU tensors (runs
anywhere).  Backend 'hip': the real gfx950 library + nccl (needs one GPU per rank).  Backend 'hipgloo': the real library, every
rank on GPU 0, gloo collectives on device tensors (a box with ONE GPU runs world 2 / 3 with real remote peers).  Every
rank trains its shards through spotlight_amd.factorization.sharded.ShardedBilinearTrainer;
rank 0 then reassembles the tables and compares them with (a) the CPU oracle and (b) the
single-device engine run on the same minibatches and negatives."""
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
from spotlight_amd.factorization.sharded import ShardedBilinearTrainer, local_rows  # noqa: E402


def main():
    backend, loss, opt, D = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    sample_on_device = len(sys.argv) > 5 and sys.argv[5] == 'sample'
    chunk_mode = len(sys.argv) > 5 and sys.argv[5] == 'chunk'  # every minibatch in ONE run_chunk call
    train_mode = len(sys.argv) > 5 and sys.argv[5] == 'train'  # bench.py's loop: trainer.train, device negatives
    slices = int(sys.argv[6]) if len(sys.argv) > 6 else None
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if backend == 'emu':
        from emu_backend import emu_lib
        dist.init_process_group('gloo')
        dev = torch.device('cpu')
        eng = _native.Engine(0, lib=emu_lib())
        stream = 0
    elif backend == 'hipgloo':
        # every rank on the ONE GPU of the box, the collectives over gloo (device tensors staged through the host by
        # ProcessGroupGloo): the real gfx950 kernels of the shard phases against a real REMOTE peer -- RCCL refuses two ranks on one
        # device, and the GPU boxes have one
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        dist.init_process_group('gloo')
        eng = _native.Engine(0)
        stream = torch.cuda.current_stream(dev).cuda_stream
    else:
        torch.cuda.set_device(rank)
        dev = torch.device('cuda', rank)
        dist.init_process_group('nccl', device_id=dev)
        eng = _native.Engine(rank)
        stream = torch.cuda.current_stream(dev).cuda_stream

    U, I, B, n_mb = 41, 53, 96, 3
    if os.environ.get('SHARD_TEST_SHAPE'):  # e.g. '300,4,1600,1': a handful of items whose lookups fill dozens of item-pass tiles
        U, I, B, n_mb = (int(x) for x in os.environ['SHARD_TEST_SHAPE'].split(','))
    N = B * n_mb - 17  # short last minibatch (or a short only one)
    rs = np.random.RandomState(123)
    users = rs.randint(0, U, N).astype(np.int64)
    items = rs.randint(0, I, N).astype(np.int64)
    # adaptive hinge (implicit.py:266-275): n_neg draws per interaction, minibatch after minibatch ONE flat draw of B * n_neg;
    # interaction j of a minibatch is scored against the flat entries [j * n_neg, (j + 1) * n_neg)
    adaptive = loss == 'adaptive_hinge'
    nn = int(os.environ.get('SHARD_TEST_NNEG', '3')) if adaptive else 1
    negs = rs.randint(0, I, N * nn).astype(np.int64)
    if train_mode:
        # every rank holds the same number of interactions of every global minibatch (as in bench.py):
        # interaction k belongs to rank k % world
        B = 24 * world
        N = B * n_mb - 2 * world  # short last minibatch, still evenly split
        users = (rs.randint(0, U // world, N) * world + np.arange(N) % world).astype(np.int64)
        items = rs.randint(0, I, N).astype(np.int64)
        negs = np.zeros(N * nn, dtype=np.int64)
    sc = min(0.3, 1.0 / np.sqrt(D))
    params = [rs.normal(0, sc, (U, D)).astype(np.float32), rs.normal(0, sc, (I, D)).astype(np.float32),
              rs.normal(0, 0.1, U).astype(np.float32), rs.normal(0, 0.1, I).astype(np.float32)]
    hp = dict(lr=0.05, weight_decay=1e-3 if opt.endswith('dense') else 0.0)

    loc = [torch.from_numpy(np.array(p[rank::world], order="C", copy=True)).to(dev) for p in params]
    assert loc[0].shape[0] == local_rows(U, world, rank) and loc[1].shape[0] == local_rows(I, world, rank)
    # adaptive hinge: a non-zero initial accumulator (torch's initial_accumulator_value).  From sum = 0 Adagrad's first step is
    # lr * sign(g) on every touched element -- 1-ulp differences in a summed gradient become O(lr) ones, and the selection (an
    # argmax over scores) then flips candidates between two runs that differ only in summation order
    acc0 = 0.1 if (adaptive and opt.startswith('adagrad')) else 0.0
    s1 = [torch.full_like(t, acc0) for t in loc]
    s2 = [torch.zeros_like(t) for t in loc]
    optim = _native.make_optim(opt, [t.data_ptr() for t in s1], [t.data_ptr() for t in s2], **hp)
    trainer = ShardedBilinearTrainer(eng, loc, optim, I, stream=stream, slices=slices)
    if sample_on_device or train_mode:
        eng.rng_set_state(np.random.RandomState(1000 + rank).get_state())
    # SHARD_TEST_SHADOW=1: the whole run inside a bias-shadow scope (this rank's item biases and their Adagrad accumulator
    # interleaved in the ctx: the owner-side gather reads the copy, the item pass updates it, the scope's end writes it back)
    shadow = trainer.bias_shadow(enabled=os.environ.get('SHARD_TEST_SHADOW') == '1')
    bias0 = loc[3].clone()
    shadow.__enter__()

    losses, used_negs = [], np.full(N * nn, -1, dtype=np.int64)
    negs2, used2 = negs.reshape(N, nn), used_negs.reshape(N, nn)  # (views: row k = the draws of interaction k)

    def mb_pos_of(idx):  # position of interaction idx inside its global minibatch
        return torch.from_numpy((np.asarray(idx) % B).astype(np.int64)).to(dev)
    kw = dict(n_neg=nn) if adaptive else {}
    use_train_loop = sample_on_device and world == 1  # ShardedBilinearTrainer.train (chunked sampling)
    if use_train_loop:
        ul = torch.from_numpy(users // world).to(dev)
        il = torch.from_numpy(items).to(dev)
        shares = trainer.train(ul, il, B, loss=loss, sample_chunk=2, **kw)
        losses = [float(x) for x in shares.cpu().numpy()]
        # the draws: one contiguous stream, minibatch after minibatch
        used_negs = np.random.RandomState(1000 + rank).randint(0, I, N * nn, dtype=np.int64)
        used2 = used_negs.reshape(N, nn)
    if train_mode:
        mine = np.nonzero(users % world == rank)[0]
        shares = trainer.train(torch.from_numpy(users[mine] // world).to(dev), torch.from_numpy(items[mine]).to(dev),
                               B // world, loss=loss, sample_chunk=2, **kw)
        dist.all_reduce(shares)
        losses = [float(x) for x in shares.cpu().numpy()]
        used2[mine] = np.random.RandomState(1000 + rank).randint(0, I, len(mine) * nn, dtype=np.int64).reshape(-1, nn)
    if chunk_mode:
        mine = np.nonzero(users % world == rank)[0]
        off = [int(np.searchsorted(mine, min(k * B, N))) for k in range(n_mb + 1)]
        gbs = [min((k + 1) * B, N) - k * B for k in range(n_mb)]
        shares = trainer.run_chunk(torch.from_numpy(users[mine] // world).to(dev), torch.from_numpy(items[mine]).to(dev),
                                   off, gbs, loss=loss, neg_in=torch.from_numpy(np.ascontiguousarray(negs2[mine]).ravel()).to(dev),
                                   mb_pos=mb_pos_of(mine) if adaptive else None, **kw)
        dist.all_reduce(shares)
        losses = [float(x) for x in shares.cpu().numpy()]
        used2[mine] = negs2[mine]
    for k in range(0 if (use_train_loop or chunk_mode or train_mode) else n_mb):
        lo, hi = k * B, min((k + 1) * B, N)
        idx = np.nonzero(users[lo:hi] % world == rank)[0] + lo
        ul = torch.from_numpy(users[idx] // world).to(dev)
        il = torch.from_numpy(items[idx]).to(dev)
        kws = dict(kw, mb_pos=mb_pos_of(idx)) if adaptive else {}
        if sample_on_device:
            neg_out = torch.full((len(idx) * nn,), -1, dtype=torch.int64, device=dev)
            part = trainer.step(ul, il, hi - lo, loss=loss, neg_out=neg_out, **kws)
            used2[idx] = neg_out.cpu().numpy().reshape(-1, nn)
        else:
            ng = torch.from_numpy(np.ascontiguousarray(negs2[idx]).ravel()).to(dev)
            part = trainer.step(ul, il, hi - lo, loss=loss, neg_in=ng, **kws)
            used2[idx] = negs2[idx]
        dist.all_reduce(part)
        losses.append(float(part.item()))
    assert optim.step == n_mb
    if os.environ.get('SHARD_TEST_SHADOW') == '1':
        assert eng.get_stat('shadowed_calls') == n_mb  # every item pass of the run indexed the copy
        assert torch.equal(loc[3], bias0)  # the caller's bias tensor is stale inside the scope: still the initial values
    shadow.__exit__(None, None, None)
    if os.environ.get('SHARD_TEST_DUMP'):  # this rank's shards after the run (tests compare two runs bit for bit)
        np.savez(os.environ['SHARD_TEST_DUMP'] + '.rank%d.npz' % rank, *[t.cpu().numpy() for t in loc + s1])

    # reassemble on rank 0
    used_negs = np.ascontiguousarray(used2).ravel()
    un = torch.from_numpy(np.where(used_negs >= 0, used_negs, 0)).to(dev)
    dist.all_reduce(un)  # every interaction belongs to exactly one rank
    used_negs = un.cpu().numpy()
    full = []
    for t, (src, st1) in enumerate(zip(loc, s1)):
        outs = []
        for tens in (src, st1):
            rows = [local_rows(params[t].shape[0], world, r) for r in range(world)]
            pad = max(rows)
            buf = torch.zeros((pad,) + tuple(tens.shape[1:]), dtype=tens.dtype, device=dev)
            buf[:tens.shape[0]] = tens
            gathered = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(gathered, buf)
            whole = np.zeros_like(params[t])
            for r in range(world):
                whole[r::world] = gathered[r][:rows[r]].cpu().numpy()
            outs.append(whole)
        full.append(outs)

    if rank == 0:
        from engine_checks import assert_close_table
        from oracle.oracle import BilinearOracle
        assert (used_negs >= 0).all() and (used_negs < I).all()
        # CPU oracle.  First minibatch: loss within 1e-5.  Whole run: hinge gradients are sums of
        # +-1/B terms that cancel to an order-dependent ~1e-9 residue which Adam's m/sqrt(v)
        # normalises to O(lr) (see engine_checks.assert_close_table), so the trajectory is judged
        # by the fraction of elements outside 1e-4 of the table norm (<= 5 %); the tight check
        # is the one against the single-device engine below.
        ora = BilinearOracle(*params, opt=opt, sparse_grads=True, state1=[np.full_like(np.asarray(x, dtype=np.float32), acc0) for x in params] if acc0 else None, **hp)
        want = ora.train(None, users, items, B, loss=loss, n_neg=nn, neg_in=used_negs)
        assert abs(losses[0] - want[0]) / abs(want[0]) < 1e-5, (losses, want)
        # (adaptive hinge: the selection is an argmax -- after the first step from zero accumulators, where every touched element
        # moves by lr * sign(g), a 1-ulp score difference flips a column's candidate and the trajectories part for good; the
        # first minibatch above and the one-GPU engine below are the checks)
        if not adaptive or acc0:
            assert np.abs(np.array(losses) - want).max() / np.abs(want).max() < 1e-3, (losses, want)
        for t in range(4 if (not adaptive or acc0) else 0):
            ref = ora.p[t].reshape(full[t][0].shape)
            bad = np.abs(full[t][0] - ref) > 1e-4 * np.abs(ref).max()
            assert bad.mean() <= 0.05, (t, bad.mean())
        # single-device engine on the same minibatches
        f = lambda a: torch.from_numpy(np.array(a, order='C', copy=True)).to(dev)
        P = [f(p) for p in params]
        S1, S2 = [torch.full_like(p, acc0) for p in P], [torch.zeros_like(p) for p in P]
        tb = _native.make_tables([p.data_ptr() for p in P], U, I, D)
        op = _native.make_optim(opt, [p.data_ptr() for p in S1], [p.data_ptr() for p in S2], **hp)
        mb = torch.zeros(n_mb, dtype=torch.float32, device=dev)
        du, di, dn = f(users), f(items), f(used_negs)
        eng.bilinear_train(tb, op, du.data_ptr(), di.data_ptr(), N, B, loss, nn, mb.data_ptr(),
                           d_neg_in=dn.data_ptr(), stream=stream)
        assert np.abs(np.array(losses) - mb.cpu().numpy()).max() / np.abs(want).max() < 1e-5
        for t in range(4):
            assert_close_table(full[t][0], P[t].cpu().numpy(), 2e-5, ('param vs single', t))
            assert_close_table(full[t][1], S1[t].cpu().numpy(), 2e-5, ('state1 vs single', t))
        print('SHARD_PARITY_OK world=%d loss=%s opt=%s D=%d' % (world, loss, opt, D))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
