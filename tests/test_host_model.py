"""Host logic of the drop-in model (spotlight_amd/factorization/implicit.py) on a GPU-less
box: the three device hooks are pointed at the fiber-emulator build of the kernels, so
fit()/predict() run end to end on CPU tensors.  Checked against the golden vectors recorded
from the live reference (same seed => same init tables, same shuffles, same negatives)."""
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import engine_checks as ec
from emu_backend import emu_lib
from oracle.replay import case_from_rec
from spotlight_amd import _native
from spotlight_amd.factorization import implicit as host
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
from spotlight_amd.factorization.representations import BilinearNet
from spotlight_amd.interactions import Interactions


@pytest.fixture()
def emu_device(monkeypatch):
    eng = _native.Engine(0, lib=emu_lib())
    monkeypatch.setattr(host, '_engine_for', lambda device: eng)
    monkeypatch.setattr(host, '_stream_for', lambda device: 0)
    monkeypatch.setattr(host, '_model_device', lambda: torch.device('cpu'))
    yield eng
    eng.close()


def _optimizer_factory(kind):
    if kind == 'adam_default':
        return None
    if kind in ('adagrad', 'adagrad_sparse'):
        return lambda params: torch.optim.Adagrad(params, lr=0.05)
    if kind == 'sparse_adam':
        return lambda params: torch.optim.SparseAdam(list(params), lr=0.01)
    if kind == 'adagrad_dense_wd':
        return lambda params: torch.optim.Adagrad(params, lr=0.05, weight_decay=1e-3)
    if kind in ('sgd', 'sgd_sparse'):
        return lambda params: torch.optim.SGD(params, lr=0.05)
    if kind == 'rmsprop':
        return lambda params: torch.optim.RMSprop(params, lr=0.01)
    raise ValueError(kind)


def _adagrad(params):
    return torch.optim.Adagrad(params, lr=0.05)


def _model_for(case):
    return ImplicitFactorizationModel(
        loss=str(case['loss']), embedding_dim=int(case['D']), n_iter=int(case['n_iter']),
        batch_size=int(case['B']), l2=float(case.get('l2', 0.0)), learning_rate=float(case.get('lr', 1e-2)),
        optimizer_func=_optimizer_factory(str(case['opt'])),
        sparse=str(case['opt']) in ('adagrad_sparse', 'sparse_adam', 'sgd_sparse'),
        random_state=np.random.RandomState(int(case['seed'])), num_negative_samples=int(case.get('n_neg', 5)))


@pytest.mark.parametrize('name', ['bpr_adam_default', 'hinge_adagrad_sparse', 'pointwise_sparse_adam',
                                  'adaptive_hinge_adagrad', 'c1_bpr_adam', 'bpr_sgd', 'adaptive_hinge_sgd_sparse',
                                  'bpr_rmsprop', 'adaptive_hinge_rmsprop'])
def test_fit_predict_match_reference_run(emu_device, name):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    inter = Interactions(rec['users'], rec['items'], num_users=int(case['U']), num_items=int(case['I']))
    model = _model_for(case)
    model._initialize(inter)
    # same torch seed + same construction order => bit-identical initial tables
    for t, w in enumerate(model._net.tables()):
        assert np.array_equal(w.detach().numpy().reshape(rec['init_%d' % t].shape), rec['init_%d' % t])
    model.fit(inter)
    st = model._random_state.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    for t, w in enumerate(model._net.tables()):
        ref = rec['final_%d' % t]
        ec.assert_open_loop_drift(w.detach().numpy().reshape(ref.shape), ref, (name, t))
    pred = model.predict(3)
    assert pred.dtype == np.float32 and pred.shape == (int(case['I']),)
    assert np.abs(pred - rec['predict_user3_all']).max() <= 2e-3 * np.abs(rec['predict_user3_all']).max()
    pairs = model.predict(rec['predict_pairs_u'], rec['predict_pairs_i'])
    assert np.abs(pairs - rec['predict_pairs']).max() <= 2e-3 * np.abs(rec['predict_pairs']).max()
    # optimizer bookkeeping stays where torch expects it
    steps = int(case['n_iter']) * ((int(case['N']) + int(case['B']) - 1) // int(case['B']))
    if 'rmsprop' in name:  # no fused update: the reference's loop through autograd, on the device
        assert model._autograd_route and model._binding is None
    elif 'sgd' not in name:  # (plain SGD keeps no step count, in torch or here)
        assert model._binding.steps_taken() == steps


def test_predict_call_forms_agree_exactly(emu_device):
    # tests/factorization/test_api.py:19-36 of the reference
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, 20, 300).astype(np.int32), rs.randint(0, 30, 300).astype(np.int32))
    model = ImplicitFactorizationModel(n_iter=1, batch_size=64, loss='bpr', random_state=np.random.RandomState(1))
    model.fit(inter)
    item_ids = np.arange(inter.num_items, dtype=np.int64)
    user_ids = np.repeat(1, inter.num_items).astype(np.int64)
    a = model.predict(1)
    b = model.predict(1, item_ids)
    c = model.predict(user_ids, item_ids)
    assert (a == b).all() and (b == c).all()
    out = model._net(torch.from_numpy(user_ids), torch.from_numpy(item_ids))
    assert np.array_equal(out.numpy(), a)


def test_resume_pickle_and_errors(emu_device):
    rs = np.random.RandomState(3)
    inter = Interactions(rs.randint(0, 15, 200).astype(np.int32), rs.randint(0, 25, 200).astype(np.int32))
    model = ImplicitFactorizationModel(n_iter=1, batch_size=32, loss='hinge',
                                       optimizer_func=_adagrad, random_state=np.random.RandomState(5))
    assert repr(model) == '<ImplicitFactorizationModel: [uninitialised]>'
    model.fit(inter)
    model.fit(inter)  # resumes: optimizer step count keeps growing
    assert model._binding.steps_taken() == 2 * 7
    assert 'BilinearNet' in repr(model)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    clone = torch.load(buf, weights_only=False)
    assert np.array_equal(clone.predict(2), model.predict(2))
    clone.fit(inter)
    with pytest.raises(ValueError, match='Maximum user id greater than number of users in model.'):
        model.predict(15)
    with pytest.raises(ValueError, match='Maximum item id greater than number of items in model.'):
        model.predict(1, np.array([25]))
    with pytest.raises(AssertionError):
        ImplicitFactorizationModel(loss='nope')
    bad = ImplicitFactorizationModel(n_iter=1, sparse=True, random_state=np.random.RandomState(1))
    with pytest.raises(RuntimeError, match='Adam does not support sparse gradients'):
        bad.fit(inter)
    # plain SGD has a fused update (dense and sparse gradients alike); momentum / RMSprop ... have none and say so
    for sparse in (False, True):
        sgd = ImplicitFactorizationModel(n_iter=2, sparse=sparse, optimizer_func=lambda p: torch.optim.SGD(p, lr=0.1),
                                         random_state=np.random.RandomState(3))
        sgd.fit(inter)
        assert np.isfinite(sgd.predict(1)).all() and not sgd._optimizer.state_dict()['state']
    # optimizers without a fused update train through the autograd route (the reference's loop on the device)
    for make in (lambda p: torch.optim.SGD(p, lr=0.1, momentum=0.9), lambda p: torch.optim.RMSprop(p, lr=0.01)):
        m2 = ImplicitFactorizationModel(n_iter=2, batch_size=50, optimizer_func=make, random_state=np.random.RandomState(3))
        m2.fit(inter)
        assert m2._autograd_route and np.isfinite(m2.predict(1)).all()
    custom = ImplicitFactorizationModel(n_iter=1, batch_size=50, loss='pointwise',
                                        representation=BilinearNet(15, 25, 16),
                                        random_state=np.random.RandomState(1))
    custom.fit(inter)
    assert custom.predict(0).shape == (25,)


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    rs = np.random.RandomState(3)
    inter = Interactions(rs.randint(0, 15, 50).astype(np.int32), rs.randint(0, 25, 50).astype(np.int32))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ImplicitFactorizationModel(n_iter=1).fit(inter)


def test_interactions_container_and_to_sequence():
    # known answers of the reference's tests/test_interactions.py:67-100
    with pytest.raises(ValueError, match='Maximum user id greater than declared number of users.'):
        Interactions(np.array([0, 5]), np.array([1, 2]), num_users=3)
    with pytest.raises(ValueError, match='Invalid ratings dimensions'):
        Interactions(np.array([0, 1]), np.array([1, 2]), ratings=np.ones(3))
    users = np.array([0] * 5 + [1] * 3, dtype=np.int32)
    items = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.int32)
    inter = Interactions(users, items, timestamps=np.arange(8))
    assert inter.tocsr().shape == (2, 9)
    seq = inter.to_sequence(max_sequence_length=5, step_size=1)
    assert (seq.sequences[:5] == np.array([[1, 2, 3, 4, 5], [0, 1, 2, 3, 4], [0, 0, 1, 2, 3],
                                           [0, 0, 0, 1, 2], [0, 0, 0, 0, 1]])).all()
    assert (seq.sequences[5:] == np.array([[0, 0, 6, 7, 8], [0, 0, 0, 6, 7], [0, 0, 0, 0, 6]])).all()
    seq2 = inter.to_sequence(max_sequence_length=5, step_size=2)
    assert (seq2.sequences[:3] == np.array([[1, 2, 3, 4, 5], [0, 0, 1, 2, 3], [0, 0, 0, 0, 1]])).all()
    assert (seq2.user_ids == np.array([0, 0, 0, 1, 1])).all()
    assert len(inter.to_sequence(5, min_sequence_length=4, step_size=1).sequences) == 2


def test_c_abi_library_exports_every_declared_symbol():
    """libspotlight_hip.so (built for gfx950 by hipcc) loads without a GPU and exports every
    entry point include/spotlight_hip.h declares; no compute is attempted."""
    import ctypes
    import re
    from spotlight_amd import build
    lib = ctypes.CDLL(build.build())
    header = open(os.path.join(os.path.dirname(GOLDEN), '..', 'include', 'spotlight_hip.h')).read()
    declared = set(re.findall(r'\b(slk_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.slk_abi_version() == _native.SLK_ABI_VERSION


# ---- BloomEmbedding layers (spotlight/layers.py:74-244) through the drop-in model API ----
def bloom_model_for(case, **kw):
    from spotlight_amd.layers import BloomEmbedding
    opt = str(case['opt'])
    model = ImplicitFactorizationModel(
        loss=str(case['loss']), embedding_dim=int(case['D']), n_iter=int(case['n_iter']),
        batch_size=int(case['B']), l2=float(case.get('l2', 0.0)), learning_rate=float(case.get('lr', 1e-2)),
        optimizer_func=_optimizer_factory(opt), random_state=np.random.RandomState(int(case['seed'])),
        num_negative_samples=int(case.get('n_neg', 5)), **kw)
    U, I, D = int(case['U']), int(case['I']), int(case['D'])
    mk = lambda n: BloomEmbedding(n, D, compression_ratio=float(case['ratio']), num_hash_functions=int(case['H']))
    ul = mk(U) if int(case['user_bloom']) else None
    il = mk(I) if int(case['item_bloom']) else None
    model._representation = BilinearNet(U, I, D, user_embedding_layer=ul, item_embedding_layer=il)
    return model


def check_bloom_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().numpy(), **kw):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    inter = Interactions(rec['users'], rec['items'], num_users=int(case['U']), num_items=int(case['I']))
    model = bloom_model_for(case, **kw)
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
    pred = model.predict(3)
    scale = np.abs(rec['predict_user3_all']).max()
    assert (np.abs(pred - rec['predict_user3_all']) > 5e-2 * scale).mean() <= 0.05
    pairs = model.predict(rec['predict_pairs_u'], rec['predict_pairs_i'])
    assert np.array_equal(pairs[:1], model.predict(int(rec['predict_pairs_u'][0]))[rec['predict_pairs_i'][:1]])
    return model


@pytest.mark.parametrize('name', ['bloom_item_bpr_adagrad', 'bloom_both_adaptive_adam',
                                  'bloom_user_pointwise_adagrad', 'bloom_c3_adaptive_adagrad'])
def test_bloom_fit_predict_match_reference_run(emu_device, name):
    model = check_bloom_fit_predict_against_fixture(name)
    assert 'BloomEmbedding' in repr(model._net)


def test_bloom_layer_api(emu_device):
    from spotlight_amd.layers import BloomEmbedding
    layer = BloomEmbedding(1000, 16, compression_ratio=0.25, num_hash_functions=3)
    assert layer.compressed_num_embeddings == 250 and layer.weight.shape == (250, 16)
    assert (layer.weight[0] == 0).all()
    with pytest.raises(ValueError):
        BloomEmbedding(10, 4, num_hash_functions=25)
    with pytest.raises(NotImplementedError):
        BloomEmbedding(10, 4, bag=True)


def check_pipelined_fit_is_value_neutral(use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """fit() of a small dataset prepares epoch e + 1's shuffle and negatives while epoch e trains
    (ImplicitFactorizationModel._fit_pipelined); the serial loop it replaces must give the same tables, losses and
    RandomState, bit for bit."""
    rs = np.random.RandomState(12)
    inter = Interactions(rs.randint(0, 90, 2000).astype(np.int32), rs.randint(0, 60, 2000).astype(np.int32),
                         num_users=90, num_items=60)
    results = []
    for limit in (host._PIPELINE_MAX_DRAWS, 0):
        old = host._PIPELINE_MAX_DRAWS
        host._PIPELINE_MAX_DRAWS = limit
        try:
            for loss, kw in (('bpr', dict(optimizer_func=_adagrad)), ('adaptive_hinge', dict(num_negative_samples=3)),
                             ('pointwise', dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=0.01)))):
                model = ImplicitFactorizationModel(loss=loss, embedding_dim=16, n_iter=3, batch_size=256, use_cuda=use_cuda,
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


def test_pipelined_fit_is_value_neutral(emu_device):
    check_pipelined_fit_is_value_neutral()


def check_prefetched_first_chunk_is_value_neutral(engine, use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """Large-epoch fit(): the next epoch's first chunk (negatives + sorts) prepared beside the last passes of this one
    (slk_bilinear_prefetch) against the same loop with the state set and the chunk prepared in line: tables, optimizer steps
    and RandomState bit for bit.  Minibatches above the persistent route's limit in chunks of two, so that every call pipelines its prep."""
    rs = np.random.RandomState(12)
    inter = Interactions(rs.randint(0, 90, 9000).astype(np.int32), rs.randint(0, 60, 9000).astype(np.int32), num_users=90, num_items=60)
    results = []
    old = host._PIPELINE_MAX_DRAWS, host._PREFETCH
    engine.set_option('chunk_interactions', 4096)
    engine.set_option('overlap_min_batch', 0)
    engine.set_option('overlap_prep', 1)  # (what _engine_for sets on the product's ctx)
    try:
        host._PIPELINE_MAX_DRAWS = 0
        for pf in (True, False):
            host._PREFETCH = pf
            before = engine.get_stat('prefetched_chunks')
            for loss, kw in (('bpr', dict(optimizer_func=_adagrad)), ('adaptive_hinge', dict(num_negative_samples=3, optimizer_func=_adagrad)),
                             ('pointwise', dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=0.01)))):
                model = ImplicitFactorizationModel(loss=loss, embedding_dim=16, n_iter=3, batch_size=2048, use_cuda=use_cuda,
                                                   random_state=np.random.RandomState(7), **kw)
                model.fit(inter)
                model.fit(inter)
                st = model._random_state.get_state()
                results.append([to_numpy(w).copy() for w in model._net.tables()] + [st[1].copy(), np.array(st[2])])
            # the route under test really ran: every epoch of every 3-epoch fit() takes over a prepared chunk
            assert engine.get_stat("prefetched_chunks") - before == (3 * 2 * 3 if pf else 0)
    finally:
        host._PIPELINE_MAX_DRAWS, host._PREFETCH = old
        engine.set_option('chunk_interactions', 1 << 23)
        engine.set_option('overlap_min_batch', 1 << 16)
    half = len(results) // 2
    for a, b in zip(results[:half], results[half:]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_prefetched_first_chunk_is_value_neutral(emu_device):
    check_prefetched_first_chunk_is_value_neutral(emu_device)


def check_bias_shadowed_fit_is_value_neutral(engine, use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """Large item tables train with {item bias, its Adagrad accumulator} interleaved for the duration of fit()
    (slk_bias_shadow_begin; host: _BIAS_SHADOW_MIN_ITEMS).  Forced on a small table, against the plain layout: tables,
    accumulators and RandomState bit for bit, over two fit() calls (the second starts from what the first wrote back) and with the
    SparseAdam model on the side, whose optimizer the shadow does not cover and whose fit() therefore never opens one."""
    rs = np.random.RandomState(13)
    inter = Interactions(rs.randint(0, 90, 12000).astype(np.int32), rs.randint(0, 60, 12000).astype(np.int32), num_users=90, num_items=60)
    results = []
    old = host._PIPELINE_MAX_DRAWS, host._BIAS_SHADOW_MIN_ITEMS
    # chunks of one minibatch, every chunk's negatives + sorts prepared on the second stream beside the passes of the chunk before
    # (fit() asks for overlap_prep = 1 itself): the shadowed passes of several chunks, overlapped
    saved = {k: engine.get_option(k) for k in ('chunk_interactions', 'overlap_min_batch')}
    engine.set_option('chunk_interactions', 4096)
    engine.set_option('overlap_min_batch', 0)
    try:
        host._PIPELINE_MAX_DRAWS = 0
        for floor in (1, 1 << 40):
            host._BIAS_SHADOW_MIN_ITEMS = floor
            before = engine.get_stat('shadowed_calls')
            for loss, kw in (('bpr', dict(optimizer_func=_adagrad)), ('adaptive_hinge', dict(num_negative_samples=3, optimizer_func=_adagrad)),
                             ('pointwise', dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=0.01)))):
                model = ImplicitFactorizationModel(loss=loss, embedding_dim=16, n_iter=2, batch_size=4096, use_cuda=use_cuda,
                                                   random_state=np.random.RandomState(7), **kw)
                model.fit(inter)
                model.fit(inter)
                st = model._random_state.get_state()
                opt_state = [to_numpy(v['sum']).copy() for v in model._optimizer.state.values() if 'sum' in v]
                results.append([to_numpy(w).copy() for w in model._net.tables()] + opt_state + [st[1].copy(), np.array(st[2])])
            # two Adagrad models x two fit() calls x two epochs (one training call each) ran on the shadow; nothing else did
            assert engine.get_stat('shadowed_calls') - before == (2 * 2 * 2 if floor == 1 else 0)
    finally:
        host._PIPELINE_MAX_DRAWS, host._BIAS_SHADOW_MIN_ITEMS = old
        for k, v in saved.items():
            engine.set_option(k, v)
    half = len(results) // 2
    for a, b in zip(results[:half], results[half:]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_bias_shadowed_fit_is_value_neutral(emu_device):
    check_bias_shadowed_fit_is_value_neutral(emu_device)


def check_user_pingponged_fit_is_value_neutral(engine, use_cuda=False, to_numpy=lambda w: w.detach().numpy()):
    """Large minibatches train on a doubled user table for the duration of fit() (slk_user_pingpong_begin; host:
    _USER_PINGPONG_MIN_BATCH).  Forced on small minibatches, against the one-table layout: tables, optimizer state, predictions and
    RandomState bit for bit, over two fit() calls (the second starts from the table the first made whole), for the optimizers the
    scope covers -- with the adaptive-hinge model on the side, whose loss it does not cover and whose fit() never opens one."""
    rs = np.random.RandomState(15)
    inter = Interactions(rs.randint(0, 90, 12000).astype(np.int32), rs.randint(0, 60, 12000).astype(np.int32), num_users=90, num_items=60)
    results = []
    old = host._PIPELINE_MAX_DRAWS, host._USER_PINGPONG_MIN_BATCH
    saved = {k: engine.get_option(k) for k in ('chunk_interactions', 'overlap_min_batch')}
    engine.set_option('chunk_interactions', 4096)
    engine.set_option('overlap_min_batch', 0)
    try:
        host._PIPELINE_MAX_DRAWS = 0
        for floor in (1, 1 << 40):
            host._USER_PINGPONG_MIN_BATCH = floor
            before = engine.get_stat('pingpong_calls')
            for loss, kw in (('bpr', dict(optimizer_func=_adagrad)), ('adaptive_hinge', dict(num_negative_samples=3, optimizer_func=_adagrad)),
                             ('pointwise', dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=0.01))),
                             ('hinge', dict(optimizer_func=lambda p: torch.optim.SGD(list(p), lr=0.05)))):
                model = ImplicitFactorizationModel(loss=loss, embedding_dim=16, n_iter=2, batch_size=4096, use_cuda=use_cuda,
                                                   random_state=np.random.RandomState(7), **kw)
                model.fit(inter)
                scores = model.predict(3).copy()  # (refused with "ping-ponged" if fit() had left the scope open)
                model.fit(inter)
                st = model._random_state.get_state()
                opt_state = [to_numpy(v[k]).copy() for v in model._optimizer.state.values() for k in ('sum', 'exp_avg', 'exp_avg_sq') if k in v]
                results.append([to_numpy(w).copy() for w in model._net.tables()] + opt_state + [scores, st[1].copy(), np.array(st[2])])
            # three covered models x two fit() calls x two epochs (one training call each) ran on the doubled table; nothing else did
            assert engine.get_stat('pingpong_calls') - before == (3 * 2 * 2 if floor == 1 else 0)
    finally:
        host._PIPELINE_MAX_DRAWS, host._USER_PINGPONG_MIN_BATCH = old
        for k, v in saved.items():
            engine.set_option(k, v)
    half = len(results) // 2
    for a, b in zip(results[:half], results[half:]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_user_pingponged_fit_is_value_neutral(emu_device):
    check_user_pingponged_fit_is_value_neutral(emu_device)


def test_exception_inside_a_pingponged_fit_makes_the_user_table_whole(emu_device, monkeypatch):
    """A fit() that dies between two epochs while its user table is doubled (slk_user_pingpong_begin) leaves on its way out what
    the one-table layout leaves: the user rows of the epochs that ran, no open scope on the shared engine."""
    rs = np.random.RandomState(16)
    inter = Interactions(rs.randint(0, 90, 12000).astype(np.int32), rs.randint(0, 60, 12000).astype(np.int32), num_users=90, num_items=60)
    old = host._PIPELINE_MAX_DRAWS, host._USER_PINGPONG_MIN_BATCH
    real_check = emu_device.check
    left = []
    try:
        host._PIPELINE_MAX_DRAWS = 0
        for floor in (1, 1 << 40):
            host._USER_PINGPONG_MIN_BATCH = floor
            model = ImplicitFactorizationModel(loss='bpr', embedding_dim=16, n_iter=3, batch_size=4096, optimizer_func=_adagrad,
                                               random_state=np.random.RandomState(7))
            calls = []

            def failing_check():
                calls.append(1)
                if len(calls) == 2:
                    raise RuntimeError('injected: epoch 1 failed')
                return real_check()
            monkeypatch.setattr(emu_device, 'check', failing_check)
            with pytest.raises(RuntimeError, match='injected'):
                model.fit(inter)
            monkeypatch.setattr(emu_device, 'check', real_check)
            scores = model.predict(3)  # (refused with "ping-ponged" if the scope were still open)
            acc = [v['sum'].numpy().copy() for v in model._optimizer.state.values() if 'sum' in v]
            left.append([w.detach().numpy().copy() for w in model._net.tables()] + acc + [scores])
            model.fit(inter)  # ... and the engine takes the next fit(), on a doubled table again or not
        for x, y in zip(left[0], left[1]):
            assert np.array_equal(x, y)
    finally:
        host._PIPELINE_MAX_DRAWS, host._USER_PINGPONG_MIN_BATCH = old


def test_fit_without_epochs_and_failed_epochs_leave_the_random_state_consistent(emu_device):
    """n_iter = 0 is a no-op on both epoch loops (the large-epoch loop raised UnboundLocalError); a degenerate epoch
    leaves the RandomState behind that epoch's negatives -- not behind the shuffle already prepared for the next one."""
    rs = np.random.RandomState(12)
    inter = Interactions(rs.randint(0, 90, 2000).astype(np.int32), rs.randint(0, 60, 2000).astype(np.int32), num_users=90, num_items=60)
    for limit in (host._PIPELINE_MAX_DRAWS, 0):
        old = host._PIPELINE_MAX_DRAWS
        host._PIPELINE_MAX_DRAWS = limit
        try:
            model = ImplicitFactorizationModel(loss='bpr', embedding_dim=8, n_iter=0, batch_size=256, optimizer_func=_adagrad,
                                               random_state=np.random.RandomState(7))
            before = model._random_state.get_state()
            model.fit(inter)
            after = model._random_state.get_state()
            assert np.array_equal(before[1], after[1]) and before[2] == after[2]
        finally:
            host._PIPELINE_MAX_DRAWS = old


def test_exception_inside_a_prefetching_fit_leaves_no_stale_prefetch(emu_device, monkeypatch):
    """the large-epoch loop prepares epoch e + 1's first chunk (slk_bilinear_prefetch) before it checks epoch e.  An
    exception between the two used to leave the prepared chunk on the shared engine, and the next training call that set no RNG
    state -- a small model's pipelined fit(), an explicit-feedback model -- was refused once (SlkError -22).  Now the loop's
    way out drops it and puts the stream back where the reference's would be."""
    rs = np.random.RandomState(12)
    inter = Interactions(rs.randint(0, 90, 9000).astype(np.int32), rs.randint(0, 60, 9000).astype(np.int32), num_users=90, num_items=60)
    small = Interactions(inter.user_ids[:600], inter.item_ids[:600], num_users=90, num_items=60)
    old = host._PIPELINE_MAX_DRAWS
    emu_device.set_option('chunk_interactions', 4096)
    emu_device.set_option('overlap_min_batch', 0)
    emu_device.set_option('overlap_prep', 1)
    try:
        host._PIPELINE_MAX_DRAWS = 0
        model = ImplicitFactorizationModel(loss='bpr', embedding_dim=16, n_iter=3, batch_size=2048, optimizer_func=_adagrad,
                                           random_state=np.random.RandomState(7))
        real_check, calls = emu_device.check, []

        def failing_check():
            calls.append(1)
            if len(calls) == 1:
                raise RuntimeError('injected: epoch 0 failed after epoch 1 was prepared ahead')
            return real_check()
        monkeypatch.setattr(emu_device, 'check', failing_check)
        with pytest.raises(RuntimeError, match='injected'):
            model.fit(inter)
        assert emu_device.get_stat('prefetch_pending') == 0
        # the RandomState is where the reference's would be: behind epoch 0's negatives
        ref = np.random.RandomState(7)
        ref.randint(-10**8, 10**8)
        ref.shuffle(np.arange(9000))
        ref.randint(0, 60, 9000, dtype=np.int64)
        got, want = model._random_state.get_state(), ref.get_state()
        assert np.array_equal(got[1], want[1]) and got[2] == want[2]
        # the next fit() on the same engine takes the pipelined route (negatives handed in: it sets no RNG state) and trains
        host._PIPELINE_MAX_DRAWS = old
        other = ImplicitFactorizationModel(loss='bpr', embedding_dim=8, n_iter=2, batch_size=256, optimizer_func=_adagrad,
                                           random_state=np.random.RandomState(3))
        other.fit(small)
        # ... and a stale chunk that does reach such a call is dropped, not refused (the C-ABI side of the same fix)
        host._PIPELINE_MAX_DRAWS = 0
        st = np.random.RandomState(5).get_state()
        model2 = ImplicitFactorizationModel(loss='bpr', embedding_dim=16, n_iter=1, batch_size=2048, optimizer_func=_adagrad,
                                            random_state=np.random.RandomState(7))
        model2.fit(inter)
        binding = model2._bind()
        d_u = torch.from_numpy(inter.user_ids.astype(np.int64))
        d_i = torch.from_numpy(inter.item_ids.astype(np.int64))
        emu_device.bilinear_prefetch(model2._slk_tables(), binding.as_struct(), d_u.data_ptr(), d_i.data_ptr(), 9000, 2048, 'bpr', 1,
                                     state=st, stream=0)
        assert emu_device.get_stat('prefetch_pending') != 0
        host._PIPELINE_MAX_DRAWS = old
        other.fit(small)
        assert emu_device.get_stat('prefetch_pending') == 0
    finally:
        host._PIPELINE_MAX_DRAWS = old
        emu_device.set_option('chunk_interactions', 1 << 23)
        emu_device.set_option('overlap_min_batch', 1 << 16)
        emu_device.set_option('overlap_prep', 0)


def test_exception_inside_a_shadowed_fit_writes_the_biases_back(emu_device, monkeypatch):
    """A fit() that dies between two epochs while its item biases are shadowed (slk_bias_shadow_begin) leaves on its way out what
    the plain layout leaves: the biases and their accumulator of the epochs that ran, no open scope on the shared engine."""
    rs = np.random.RandomState(14)
    inter = Interactions(rs.randint(0, 90, 12000).astype(np.int32), rs.randint(0, 60, 12000).astype(np.int32), num_users=90, num_items=60)
    old = host._PIPELINE_MAX_DRAWS, host._BIAS_SHADOW_MIN_ITEMS
    real_check = emu_device.check
    left = []
    try:
        host._PIPELINE_MAX_DRAWS = 0
        for floor in (1, 1 << 40):
            host._BIAS_SHADOW_MIN_ITEMS = floor
            model = ImplicitFactorizationModel(loss='bpr', embedding_dim=16, n_iter=3, batch_size=4096, optimizer_func=_adagrad,
                                               random_state=np.random.RandomState(7))
            calls = []

            def failing_check():
                calls.append(1)
                if len(calls) == 2:
                    raise RuntimeError('injected: epoch 1 failed')
                return real_check()
            monkeypatch.setattr(emu_device, 'check', failing_check)
            with pytest.raises(RuntimeError, match='injected'):
                model.fit(inter)
            monkeypatch.setattr(emu_device, 'check', real_check)
            scores = model.predict(3)  # (refused with "shadowed" if the scope were still open)
            acc = [v['sum'].numpy().copy() for v in model._optimizer.state.values() if 'sum' in v]
            left.append([w.detach().numpy().copy() for w in model._net.tables()] + acc + [scores])
            model.fit(inter)  # ... and the engine takes the next fit(), shadowed again or not
        assert np.abs(left[0][3]).max() > 0  # the item biases did train
        for x, y in zip(left[0], left[1]):
            assert np.array_equal(x, y)
    finally:
        host._PIPELINE_MAX_DRAWS, host._BIAS_SHADOW_MIN_ITEMS = old


def test_id_checks_on_worker_threads_raise_like_the_serial_ones(emu_device):
    """Large fits run the id-range checks (implicit.py:169-182) on worker threads beside the upload and the first shuffle: the
    same exceptions in the same order, before anything is trained, RandomState and tables as they were."""
    rs = np.random.RandomState(3)
    users, items = rs.randint(0, 90, 3000).astype(np.int32), rs.randint(0, 60, 3000).astype(np.int32)
    good = Interactions(users, items, num_users=90, num_items=60)
    old = host._PIPELINE_MAX_DRAWS, host._DEFERRED_CHECK_MIN
    host._PIPELINE_MAX_DRAWS, host._DEFERRED_CHECK_MIN = 0, 1
    try:
        model = ImplicitFactorizationModel(loss='bpr', embedding_dim=8, n_iter=2, batch_size=512, optimizer_func=_adagrad,
                                           random_state=np.random.RandomState(7))
        model.fit(good)
        for bad_users, bad_items, exc, text in ((users + 1000, items, ValueError, 'user id'), (users, items + 1000, ValueError, 'item id'),
                                               (users + 1000, items + 1000, ValueError, 'user id'), (users, items - 70, IndexError, 'index out of range')):
            before = model._random_state.get_state()
            tables = [t.detach().cpu().numpy().copy() for t in model._net.tables()]
            with pytest.raises(exc, match=text):
                model.fit(Interactions(bad_users, bad_items, num_users=90, num_items=60))
            after = model._random_state.get_state()
            assert np.array_equal(before[1], after[1]) and before[2] == after[2]
            for a, t in zip(tables, model._net.tables()):
                assert np.array_equal(a, t.detach().cpu().numpy())
        model.fit(good)  # and the model still trains
    finally:
        host._PIPELINE_MAX_DRAWS, host._DEFERRED_CHECK_MIN = old


class _TwoTower(torch.nn.Module):
    """A representation that is NOT BilinearNet (the reference accepts any module with forward(user_ids, item_ids),
    spotlight/factorization/implicit.py:131-139): a tanh layer on the user side, this package's embedding layers below it."""

    def __init__(self, num_users, num_items, dim=8):
        super(_TwoTower, self).__init__()
        from spotlight_amd.layers import ScaledEmbedding, ZeroEmbedding
        self.user_embeddings = ScaledEmbedding(num_users, dim)
        self.item_embeddings = ScaledEmbedding(num_items, dim)
        self.item_biases = ZeroEmbedding(num_items, 1)
        self.mix = torch.nn.Linear(dim, dim)

    def forward(self, user_ids, item_ids):
        u = torch.tanh(self.mix(self.user_embeddings(user_ids)))
        return (u * self.item_embeddings(item_ids)).sum(1) + self.item_biases(item_ids).squeeze()


@pytest.mark.parametrize('loss', ['bpr', 'adaptive_hinge'])
def test_custom_representation_trains_through_the_autograd_route(emu_device, loss):
    rs = np.random.RandomState(0)
    users = rs.randint(0, 15, 400).astype(np.int32)
    items = ((users * 3 + rs.randint(0, 2, 400)) % 25).astype(np.int32)  # learnable structure
    inter = Interactions(users, items, num_users=15, num_items=25)
    torch.manual_seed(0)
    model = ImplicitFactorizationModel(loss=loss, n_iter=1, batch_size=64, learning_rate=5e-2, representation=_TwoTower(15, 25),
                                       random_state=np.random.RandomState(2), num_negative_samples=3)
    before = np.random.RandomState(2)
    before.randint(-10**8, 10**8)  # the constructor's draw
    model.fit(inter)
    assert model._autograd_route and model._binding is None
    # the RandomState was consumed as the reference consumes it: one shuffle + one randint per minibatch, on the host
    order = np.arange(400)
    before.shuffle(order)
    for lo in range(0, 400, 64):
        before.randint(0, 25, (min(64, 400 - lo)) * (3 if loss == 'adaptive_hinge' else 1), dtype=np.int64)
    assert (model._random_state.get_state()[1] == before.get_state()[1]).all()
    first = model.predict(1)
    model._n_iter = 15
    model.fit(inter)
    pred = model.predict(1)
    assert pred.shape == (25,) and pred.dtype == np.float32 and np.isfinite(pred).all()
    pos = np.unique(items[users == 1])
    neg = np.setdiff1d(np.arange(25), pos)
    assert pred[pos].mean() > pred[neg].mean() and not np.array_equal(first, pred)
    assert np.array_equal(model.predict(np.array([1, 1]), np.array([3, 4])), pred[[3, 4]])
