"""Engine sources (spotlight_amd/csrc/*.hip, unmodified) executed by the fiber emulator and
compared with the CPU oracle / numpy.  Covers what a GPU-less box can cover: indices, keys,
segments, loss/optimizer formulas and the C-ABI control flow."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from emu_backend import HostModel, emu_lib, ptr
from oracle.oracle import BilinearOracle, Rng
from oracle.replay import ORACLE_OPT, case_from_rec, oracle_hparams
from spotlight_amd import _native


@pytest.fixture(scope='module')
def eng():
    e = _native.Engine(0, lib=emu_lib())
    yield e
    e.close()


@pytest.mark.parametrize('num_items', [1, 2, 100, 1682, 4096, 10 ** 6, 10 ** 9, 2 ** 32])
def test_sampler_bit_exact(eng, num_items):
    rs = np.random.RandomState(1234)
    rs.randint(0, 10, 77)  # start mid-block
    eng.rng_set_state(rs.get_state())
    for count in (1, 5, 700, 3000):
        out = np.full(count, -1, dtype=np.int64)
        eng.sample_items(num_items, count, ptr(out))
        want = rs.randint(0, num_items, count, dtype=np.int64)
        assert (out == want).all()
        got, ref = eng.rng_get_state(), rs.get_state()
        assert (got[1] == ref[1]).all() and got[2] == ref[2]


def test_sampler_fresh_seed_and_block_boundaries(eng):
    rs = np.random.RandomState(42)  # pos == 624: regenerate-on-first-draw
    eng.rng_set_state(rs.get_state())
    for count in (624, 1, 623, 1248):
        out = np.empty(count, dtype=np.int64)
        eng.sample_items(2 ** 32, count, ptr(out))  # every word accepted: lands on block edges
        assert (out == rs.randint(0, 2 ** 32, count, dtype=np.int64)).all()
        got, ref = eng.rng_get_state(), rs.get_state()
        assert (got[1] == ref[1]).all() and got[2] == ref[2]


def _rel_inf(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CASES = [(loss, opt) for loss in ('pointwise', 'bpr', 'hinge', 'adaptive_hinge')
         for opt in ('adagrad', 'sparse_adam', 'adam_dense', 'adagrad_dense')]


@pytest.mark.parametrize('loss,opt', CASES)
@pytest.mark.parametrize('D', [8, 6])
def test_train_matches_oracle(eng, loss, opt, D):
    rs = np.random.RandomState(5)
    U, I, N, B, nn = 37, 29, 150, 64, 3
    users = rs.randint(0, U, N).astype(np.int64)
    items = rs.randint(0, I, N).astype(np.int64)
    params = [rs.normal(0, 0.3, (U, D)), rs.normal(0, 0.3, (I, D)), rs.normal(0, 0.1, U), rs.normal(0, 0.1, I)]
    hp = dict(lr=0.05, weight_decay=1e-3 if opt.endswith('dense') else 0.0)
    ora = BilinearOracle(*params, opt=opt, sparse_grads=True, **hp)
    dev = HostModel(params, opt=opt, **hp)
    state = np.random.RandomState(9).get_state()
    orng = Rng(state=state)
    eng.rng_set_state(state)
    n_mb = (N + B - 1) // B
    for epoch in range(2):
        want_loss, want_neg = ora.train(orng, users, items, B, loss=loss, n_neg=nn, want_negs=True)
        mb_loss = np.zeros(n_mb, dtype=np.float32)
        neg_out = np.full(want_neg.size, -1, dtype=np.int64)
        eng.bilinear_train(dev.tables, dev.optim, ptr(users), ptr(items), N, B, loss, nn, ptr(mb_loss),
                           d_neg_out=ptr(neg_out))
        assert (neg_out == want_neg).all()
        assert np.abs(mb_loss - want_loss).max() / np.abs(want_loss).max() < 1e-5
    assert dev.optim.step == ora.step_count == 2 * n_mb
    for t in range(4):
        assert _rel_inf(dev.p[t], ora.p[t]) < 2e-5, t
        assert _rel_inf(dev.s1[t], ora.s1[t]) < 2e-5, t
        if opt in ('sparse_adam', 'adam_dense'):
            assert _rel_inf(dev.s2[t], ora.s2[t]) < 2e-5, t
    got, ref = eng.rng_get_state(), orng.get_state()
    assert (got[1] == ref[1]).all() and got[2] == ref[2]
    # predict: scalar user vs all items, and explicit pairs
    out = np.empty(I, dtype=np.float32)
    u3 = np.array([3], dtype=np.int64)
    eng.bilinear_predict(dev.tables, ptr(u3), 1, None, I, ptr(out))
    assert _rel_inf(out, ora.predict(3)) < 1e-5
    pu, pi = users[:50].copy(), items[:50].copy()
    out = np.empty(50, dtype=np.float32)
    eng.bilinear_predict(dev.tables, ptr(pu), 50, ptr(pi), 50, ptr(out))
    assert _rel_inf(out, ora.predict(pu, pi)) < 1e-5


@pytest.mark.parametrize('name', ['bpr_adagrad_sparse', 'hinge_sparse_adam', 'pointwise_adam_default',
                                  'adaptive_hinge_adagrad', 'd64_bpr_adagrad', 'c1_bpr_adam',
                                  'd12_pointwise_adagrad_wd'])
def test_train_replays_reference_fixture(eng, name):
    """Golden vectors recorded from the live reference: same shuffled ids, same seed ->
    bit-exact negatives, losses within 1e-5 (first minibatch) / 1e-4 (trajectory)."""
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    hp = oracle_hparams(case)
    hp.pop('sparse_grads')
    dev = HostModel([rec['init_%d' % t] for t in range(4)], opt=ORACLE_OPT[case['opt']], **hp)
    eng.rng_set_state(('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    host = Rng(state=('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    nn = int(case.get('n_neg', 5)) if case['loss'] == 'adaptive_hinge' else 1
    N, B = int(case['N']), int(case['B'])
    n_mb = (N + B - 1) // B
    losses, negs = [], []
    for e in range(int(case['n_iter'])):
        # the shuffle stays on the host (torch_utils.py:35-52) and shares the stream
        host.set_state(eng.rng_get_state())
        perm = host.shuffle_perm(N)
        eng.rng_set_state(host.get_state())
        su = rec['users'].astype(np.int64)[perm]
        si = rec['items'].astype(np.int64)[perm]
        assert (su == rec['shuffled_users'][e]).all()
        mb_loss = np.zeros(n_mb, dtype=np.float32)
        neg_out = np.empty(N * nn, dtype=np.int64)
        eng.bilinear_train(dev.tables, dev.optim, ptr(su), ptr(si), N, B, case['loss'], nn, ptr(mb_loss),
                           d_neg_out=ptr(neg_out))
        losses.append(mb_loss)
        negs.append(neg_out)
    assert (np.concatenate(negs) == rec['negatives']).all()
    losses = np.concatenate(losses)
    assert abs(losses[0] - rec['losses'][0]) / abs(rec['losses'][0]) < 1e-5
    # later minibatches: trajectories are only conditionally stable (see oracle/make_golden.py:
    # Adagrad's first step is lr*g/(|g|+1e-10), and at init bpr gradients of an item that is
    # positive in one interaction and negative in another cancel to ~1e-11, so the update
    # depends on summation order -- torch's own dense and sparse paths differ by this much)
    assert np.max(np.abs(losses - rec['losses']) / np.abs(rec['losses'])) < 1e-3
    st = eng.rng_get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    for t in range(4):
        ref = rec['final_%d' % t]
        bad = np.abs(dev.p[t].reshape(ref.shape) - ref) > 1e-3 * np.abs(ref).max()
        assert bad.mean() <= 0.05, (t, bad.mean())


def test_argument_errors(eng):
    dev = HostModel([np.zeros((4, 8)), np.zeros((5, 8)), np.zeros(4), np.zeros(5)])
    ids = np.zeros(4, dtype=np.int64)
    loss = np.zeros(1, dtype=np.float32)
    with pytest.raises(_native.SlkError) as e:
        eng.bilinear_train(dev.tables, dev.optim, ptr(ids), ptr(ids), 4, 0, 'bpr', 1, ptr(loss))
    assert e.value.code == _native.SLK_EINVAL
    bad = _native.make_tables([ptr(x) for x in dev.p], 4, 5, 67)  # dim 67: not %4, > 64
    with pytest.raises(_native.SlkError):
        eng.bilinear_train(bad, dev.optim, ptr(ids), ptr(ids), 4, 2, 'bpr', 1, ptr(loss))
    with pytest.raises(_native.SlkError):
        eng.sample_items(0, 4, ptr(ids))
