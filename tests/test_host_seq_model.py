"""Host logic of the drop-in sequence model (spotlight_amd/sequence/implicit.py) on a GPU-less
box through the emulator build of the kernels, against the sequence fixtures recorded from the
live reference (same seed => same init tables, same composed shuffles, same negatives)."""
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from emu_backend import emu_lib
from oracle.replay import case_from_rec
from spotlight_amd import _native
from spotlight_amd.factorization import implicit as host
from spotlight_amd.interactions import Interactions, SequenceInteractions
from spotlight_amd.sequence.implicit import ImplicitSequenceModel
from spotlight_amd.sequence.representations import PoolNet


@pytest.fixture()
def emu_device(monkeypatch):
    eng = _native.Engine(0, lib=emu_lib())
    monkeypatch.setattr(host, '_engine_for', lambda device: eng)
    monkeypatch.setattr(host, '_stream_for', lambda device: 0)
    monkeypatch.setattr(host, '_model_device', lambda: torch.device('cpu'))
    yield eng
    eng.close()


def _adagrad(params):
    return torch.optim.Adagrad(params, lr=0.05)


def _sparse_adam(params):
    return torch.optim.SparseAdam(list(params), lr=0.01)


def model_for(case, **kw):
    opt = str(case['opt'])
    of = {'adam_default': None, 'adagrad': _adagrad, 'adagrad_sparse': _adagrad, 'sparse_adam': _sparse_adam}[opt]
    representation = 'pooling'
    if int(case.get('bloom', 0)):
        # as recorded by oracle/make_golden_seq.py: the net is built by the caller, after torch.manual_seed
        from spotlight_amd.layers import BloomEmbedding
        torch.manual_seed(int(case['seed']))
        representation = PoolNet(int(case['I']), int(case['D']), item_embedding_layer=BloomEmbedding(
            int(case['I']), int(case['D']), compression_ratio=float(case['ratio']),
            num_hash_functions=int(case['bloom']), padding_idx=0))
    return ImplicitSequenceModel(
        loss=str(case['loss']), representation=representation, embedding_dim=int(case['D']),
        n_iter=int(case['n_iter']), batch_size=int(case['B']), l2=float(case.get('l2', 0.0)),
        learning_rate=float(case.get('lr', 1e-2)), optimizer_func=of,
        sparse=opt in ('adagrad_sparse', 'sparse_adam'), random_state=np.random.RandomState(int(case['seed'])),
        num_negative_samples=int(case.get('n_neg', 5)), **kw)


def check_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().numpy(), **kw):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    inter = SequenceInteractions(rec['sequences'], num_items=int(case['I']))
    model = model_for(case, **kw)
    model._initialize(inter)
    for t, w in enumerate(model._net.tables()):
        assert np.array_equal(to_numpy(w).reshape(rec['init_%d' % t].shape), rec['init_%d' % t])
    model.fit(inter)
    st = model._random_state.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    for t, w in enumerate(model._net.tables()):
        ref = rec['final_%d' % t]
        bad = np.abs(to_numpy(w).reshape(ref.shape) - ref) > 1e-3 * np.abs(ref).max()
        assert bad.mean() <= max(0.05, float(case.get('frac_tol', 0.05)))
    pred = model.predict(rec['predict_seq'])
    assert pred.dtype == np.float32 and pred.shape == (int(case['I']),)
    # trajectories drift on the ill-conditioned elements (see oracle/make_golden_seq.py: a bias whose
    # gradient is +-1/M cancellation noise moves by O(lr) under Adam/Adagrad), so predictions are
    # judged like the tables: by the fraction of items outside tolerance
    scale = np.abs(rec['predict_all']).max()
    bad = np.abs(pred - rec['predict_all']) > 5e-2 * scale
    assert bad.mean() <= max(0.05, float(case.get('frac_tol', 0.05))), bad.mean()
    some = model.predict(rec['predict_seq2'], rec['predict_items'])
    assert some.shape == (rec['predict_items'].size,)
    # the three call forms agree with each other exactly (tests/sequence of the reference)
    assert np.array_equal(model.predict(rec['predict_seq2'])[rec['predict_items'].ravel()], some)
    return model


@pytest.mark.parametrize('name', ['seq_bpr_adam_default', 'seq_hinge_adagrad_sparse', 'seq_pointwise_sparse_adam',
                                  'seq_adaptive_hinge_adagrad', 'seq_d64_bpr_adagrad',
                                  'seq_bloom_bpr_adagrad', 'seq_bloom_pointwise_adam_default',
                                  'seq_bloom_d64_bpr_adagrad'])
def test_fit_predict_match_reference_run(emu_device, name):
    model = check_fit_predict_against_fixture(name)
    st = model._optimizer.state[model._net.item_embeddings.weight]
    assert int(torch.as_tensor(st['step']).item()) > 0


def test_resume_and_pickle(emu_device):
    rs = np.random.RandomState(0)
    seqs = rs.randint(1, 30, (40, 6)).astype(np.int32)
    inter = SequenceInteractions(seqs, num_items=30)
    mk = lambda n_iter: ImplicitSequenceModel(loss='bpr', embedding_dim=8, n_iter=n_iter, batch_size=16,
                                              optimizer_func=_adagrad, random_state=np.random.RandomState(5))
    a = mk(2)
    a.fit(inter)
    b = mk(1)
    b.fit(inter)
    buf = io.BytesIO()
    torch.save(b, buf)
    buf.seek(0)
    b = torch.load(buf, weights_only=False)
    b.fit(inter)  # resumes: parameters, optimizer state, RandomState and the composed shuffle order
    # (the composed shuffle of the reference restarts from the caller's array on every fit(), so
    # two fit(n_iter=1) calls differ from one fit(n_iter=2) only through the data order)
    assert np.isfinite(b.predict(seqs[0])).all()
    assert a._net.tables()[0].shape == b._net.tables()[0].shape


def test_errors_and_validation(emu_device):
    seqs = np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int32)
    inter = SequenceInteractions(seqs, num_items=6)
    with pytest.raises(AssertionError):
        ImplicitSequenceModel(loss='nope')
    with pytest.raises(AssertionError):
        ImplicitSequenceModel(representation='transformer')
    lstm = ImplicitSequenceModel(representation='lstm', n_iter=1, embedding_dim=8, random_state=np.random.RandomState(1))
    lstm.fit(inter)  # torch-side encoder over the embedding front-end (tests/test_host_encoders.py)
    assert lstm.predict(seqs[0]).shape == (6,)
    m = ImplicitSequenceModel(loss='bpr', n_iter=1, embedding_dim=8, random_state=np.random.RandomState(1))
    m.fit(inter)
    with pytest.raises(ValueError):
        m.predict(np.array([1, 9]))
    with pytest.raises(ValueError):
        m.fit(SequenceInteractions(np.array([[7, 1, 2]], dtype=np.int32), num_items=6))
    # a custom PoolNet is accepted as the representation (sequence/implicit.py:158-159)
    net = PoolNet(6, 8)
    m2 = ImplicitSequenceModel(loss='hinge', representation=net, n_iter=1, random_state=np.random.RandomState(1))
    m2.fit(inter)
    assert m2._net is net and (net.item_embeddings.weight[0] == 0).all()


def test_to_sequence_feeds_the_model(emu_device):
    rs = np.random.RandomState(3)
    n = 400
    inter = Interactions(rs.randint(0, 20, n).astype(np.int32), rs.randint(1, 50, n).astype(np.int32),
                         timestamps=rs.randint(0, 1000, n).astype(np.int32), num_users=20, num_items=50)
    seq = inter.to_sequence(max_sequence_length=8, min_sequence_length=2, step_size=3)
    m = ImplicitSequenceModel(loss='pointwise', n_iter=2, embedding_dim=8, batch_size=32,
                              random_state=np.random.RandomState(2))
    m.fit(seq)
    assert m.predict(seq.sequences[0]).shape == (50,)


def test_to_sequence_device_route(emu_device):
    """Interactions.to_sequence(device='cuda') (slk_seqprep.hip) returns what the host route returns -- on the
    reference-recorded split of tests/golden/host_api.npz, on the worked example of the reference's own test
    (tests/test_interactions.py:67-100: every window ends where the docstring says) and with the argument
    forms the reference accepts."""
    import os

    from conftest import GOLDEN
    from spotlight_amd.cross_validation import random_train_test_split
    from spotlight_amd.datasets.synthetic import generate_sequential
    rec = np.load(os.path.join(GOLDEN, 'host_api.npz'))
    data = generate_sequential(num_users=30, num_items=60, num_interactions=900, concentration_parameter=0.1, order=3,
                               random_state=np.random.RandomState(11))
    _, te = random_train_test_split(data, test_percentage=0.25, random_state=np.random.RandomState(13))
    seq = te.to_sequence(max_sequence_length=6, min_sequence_length=2, step_size=2, device='cuda')
    assert np.array_equal(seq.sequences, rec['to_seq']) and seq.sequences.dtype == np.int32
    assert seq.num_items == te.num_items and seq.max_sequence_length == 6

    rs = np.random.RandomState(5)
    n = 700
    inter = Interactions(rs.randint(0, 25, n).astype(np.int32), rs.randint(1, 80, n).astype(np.int32),
                         timestamps=rs.randint(0, 300, n).astype(np.int32), num_users=25, num_items=80)
    for kw in (dict(), dict(max_sequence_length=4), dict(max_sequence_length=5, step_size=1),
               dict(max_sequence_length=7, min_sequence_length=3, step_size=2), dict(max_sequence_length=3, min_sequence_length=0)):
        a, b = inter.to_sequence(**kw), inter.to_sequence(device='cuda', **kw)
        assert np.array_equal(a.sequences, b.sequences) and np.array_equal(a.user_ids, b.user_ids), kw
        assert b.user_ids.dtype == np.int32
    with pytest.raises(IndexError):
        inter.to_sequence(max_sequence_length=4, min_sequence_length=5, device='cuda')
    with pytest.raises(ValueError):
        inter.to_sequence(device='cpu')
    no_ts = Interactions(inter.user_ids, inter.item_ids)
    with pytest.raises(ValueError):
        no_ts.to_sequence(device='cuda')
    m = ImplicitSequenceModel(loss='bpr', n_iter=1, embedding_dim=8, batch_size=32, random_state=np.random.RandomState(2))
    m.fit(inter.to_sequence(max_sequence_length=8, min_sequence_length=2, step_size=3, device='cuda'))


def check_pipelined_seq_fit_is_value_neutral(use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """ImplicitSequenceModel._fit_pipelined (next epoch's composed shuffle + negatives drawn while this epoch trains) against
    the serial epoch loop: tables and RandomState bit for bit."""
    from spotlight_amd.factorization import implicit as host_mod
    from spotlight_amd.interactions import SequenceInteractions
    rs = np.random.RandomState(5)
    seqs = rs.randint(1, 80, (150, 12)).astype(np.int32)
    seqs[rs.rand(150) < 0.4, :4] = 0
    data = SequenceInteractions(seqs, num_items=80)
    results = []
    for limit in (host_mod._PIPELINE_MAX_DRAWS, 0):
        old = host_mod._PIPELINE_MAX_DRAWS
        host_mod._PIPELINE_MAX_DRAWS = limit
        try:
            for loss, kw in (('bpr', {}), ('adaptive_hinge', dict(num_negative_samples=3))):
                model = ImplicitSequenceModel(loss=loss, representation='pooling', embedding_dim=16, n_iter=3, batch_size=64,
                                              use_cuda=use_cuda, random_state=np.random.RandomState(3), **kw)
                model.fit(data)
                model.fit(data)
                st = model._random_state.get_state()
                results.append([to_numpy(w).copy() for w in model._net.tables()] + [st[1].copy(), np.array(st[2])])
        finally:
            host_mod._PIPELINE_MAX_DRAWS = old
    half = len(results) // 2
    for a, b in zip(results[:half], results[half:]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_pipelined_seq_fit_is_value_neutral(emu_device):
    check_pipelined_seq_fit_is_value_neutral()
