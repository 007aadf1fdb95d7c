"""Host logic of the drop-in explicit-feedback model (spotlight_amd/factorization/explicit.py) through the
emulator build of the kernels, against the fixtures recorded from the live reference's
ExplicitFactorizationModel (same seed => same init tables, same shuffles)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from emu_backend import emu_lib
from oracle.replay import case_from_rec
from spotlight_amd import _native
from spotlight_amd import evaluation as ev
from spotlight_amd.factorization import implicit as host
from spotlight_amd.factorization.explicit import ExplicitFactorizationModel
from spotlight_amd.interactions import Interactions


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


def check_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().numpy(), **kw):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    opt = str(case['opt'])
    of = {'adam_default': None, 'adagrad': _adagrad, 'adagrad_sparse': _adagrad, 'sparse_adam': _sparse_adam}[opt]
    inter = Interactions(rec['users'], rec['items'], ratings=rec['ratings'], num_users=int(case['U']),
                         num_items=int(case['I']))
    model = ExplicitFactorizationModel(
        loss=str(case['loss']), embedding_dim=int(case['D']), n_iter=int(case['n_iter']), batch_size=int(case['B']),
        l2=float(case.get('l2', 0.0)), learning_rate=float(case.get('lr', 1e-2)), optimizer_func=of,
        sparse=opt in ('adagrad_sparse', 'sparse_adam'), random_state=np.random.RandomState(int(case['seed'])), **kw)
    model._initialize(inter)
    for t, w in enumerate(model._net.tables()):
        assert np.array_equal(to_numpy(w).reshape(rec['init_%d' % t].shape), rec['init_%d' % t])
    model.fit(inter)
    st = model._random_state.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    for t, w in enumerate(model._net.tables()):
        ref = rec['final_%d' % t]
        bad = np.abs(to_numpy(w).reshape(ref.shape) - ref) > 1e-3 * np.abs(ref).max()
        assert bad.mean() <= max(0.05, float(case.get('frac_tol', 0.05))), (t, bad.mean())
    pred = model.predict(3)
    assert pred.dtype == np.float32 and pred.shape == (int(case['I']),)
    assert np.abs(pred - rec['predict_all']).max() <= 2e-3 * np.abs(rec['predict_all']).max()
    pairs = model.predict(rec['predict_users'], rec['predict_items'])
    assert np.abs(pairs - rec['predict_pairs']).max() <= 2e-3 * np.abs(rec['predict_pairs']).max()
    assert np.isfinite(ev.rmse_score(model, inter))
    return model


@pytest.mark.parametrize('name', ['explicit_regression_adam_default', 'explicit_poisson_adagrad',
                                  'explicit_logistic_sparse_adam', 'explicit_d64_regression_adagrad'])
def test_fit_predict_match_reference_run(emu_device, name):
    check_fit_predict_against_fixture(name)


def check_pipelined_explicit_fit_is_value_neutral(use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """fit() of a small dataset shuffles epoch e + 1 while epoch e trains (ExplicitFactorizationModel._fit_pipelined); the
    serial loop it replaces must give the same tables and RandomState, bit for bit."""
    rs = np.random.RandomState(12)
    inter = Interactions(rs.randint(0, 90, 2000).astype(np.int32), rs.randint(0, 60, 2000).astype(np.int32),
                         ratings=rs.randint(1, 6, 2000).astype(np.float32), num_users=90, num_items=60)
    results = []
    for limit in (host._PIPELINE_MAX_DRAWS, 0):
        old = host._PIPELINE_MAX_DRAWS
        host._PIPELINE_MAX_DRAWS = limit
        try:
            for loss, kw in (('regression', dict(optimizer_func=_adagrad)), ('poisson', dict()),
                             ('logistic', dict(sparse=True, optimizer_func=_sparse_adam))):
                model = ExplicitFactorizationModel(loss=loss, embedding_dim=16, n_iter=3, batch_size=256, use_cuda=use_cuda,
                                                   random_state=np.random.RandomState(7), **kw)
                model.fit(inter)
                model.fit(inter)  # resume: optimizer steps and the RandomState carry over
                st = model._random_state.get_state()
                results.append([to_numpy(w).copy() for w in model._net.tables()] + [st[1].copy(), np.array(st[2])])
        finally:
            host._PIPELINE_MAX_DRAWS = old
    half = len(results) // 2
    for a, b in zip(results[:half], results[half:]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_pipelined_explicit_fit_is_value_neutral(emu_device):
    check_pipelined_explicit_fit_is_value_neutral()


def test_errors(emu_device):
    with pytest.raises(AssertionError):
        ExplicitFactorizationModel(loss='bpr')
    inter = Interactions(np.array([0, 1], dtype=np.int32), np.array([1, 2], dtype=np.int32), num_users=3, num_items=4)
    m = ExplicitFactorizationModel(n_iter=1, embedding_dim=8, random_state=np.random.RandomState(1))
    with pytest.raises(TypeError):
        m.fit(inter)  # no ratings


def test_mrr_score_on_an_explicit_model_takes_the_predict_route(emu_device):
    """evaluation.mrr_score(explicit_model, test) works as in the reference: ranking of predict()'s
    ratings (no device fast path for explicit models)."""
    model = check_fit_predict_against_fixture('explicit_poisson_adagrad')
    rs = np.random.RandomState(2)
    test = Interactions(rs.randint(0, model._num_users, 60).astype(np.int32),
                        rs.randint(0, model._num_items, 60).astype(np.int32),
                        num_users=model._num_users, num_items=model._num_items)
    got = ev.mrr_score(model, test)

    class OnlyPredict(object):
        def predict(self, *a, **kw):
            return model.predict(*a, **kw)
    want = ev.mrr_score(OnlyPredict(), test)
    assert got.shape == want.shape and np.array_equal(got, want)
