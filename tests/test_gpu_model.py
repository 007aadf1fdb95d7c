"""`-m gpu`: the drop-in model API on cuda:0 (real library), against the golden vectors
recorded from the live reference, plus size-independent properties at larger sizes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.replay import case_from_rec

pytestmark = pytest.mark.gpu


def _adagrad(params):
    return torch.optim.Adagrad(params, lr=0.05)


def _sparse_adam(params):
    return torch.optim.SparseAdam(list(params), lr=0.01)


def _model_for(case):
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    opt = str(case['opt'])
    of = {'adam_default': None, 'adagrad': _adagrad, 'adagrad_sparse': _adagrad, 'sparse_adam': _sparse_adam}[opt]
    return ImplicitFactorizationModel(
        loss=str(case['loss']), embedding_dim=int(case['D']), n_iter=int(case['n_iter']),
        batch_size=int(case['B']), l2=float(case.get('l2', 0.0)), learning_rate=float(case.get('lr', 1e-2)),
        optimizer_func=of, sparse=opt in ('adagrad_sparse', 'sparse_adam'), use_cuda=True,
        random_state=np.random.RandomState(int(case['seed'])), num_negative_samples=int(case.get('n_neg', 5)))


@pytest.mark.parametrize('name', ['bpr_adam_default', 'hinge_adagrad_sparse', 'pointwise_sparse_adam',
                                  'adaptive_hinge_adagrad', 'adaptive_hinge_sparse_adam', 'c1_bpr_adam',
                                  'd64_adaptive_sparse_adam'])
def test_fit_predict_match_reference_run(name):
    from spotlight_amd.interactions import Interactions
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    inter = Interactions(rec['users'], rec['items'], num_users=int(case['U']), num_items=int(case['I']))
    model = _model_for(case)
    model._initialize(inter)
    for t, w in enumerate(model._net.tables()):
        assert w.is_cuda
        assert np.array_equal(w.detach().cpu().numpy().reshape(rec['init_%d' % t].shape), rec['init_%d' % t])
    model.fit(inter)
    st = model._random_state.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    # fit() must land exactly where the engine-level replay of the same recording lands (the same kernels driven through
    # the C ABI directly; engine_checks.check_replays_reference_fixture validates that replay step by step against the
    # oracle): bit-identical tables, no tolerance
    import engine_checks as ec
    from hip_backend import HipBackend
    be = HipBackend()
    try:
        want = ec.check_replays_reference_fixture(be, GOLDEN, name)
    finally:
        be.close()
    for t, w in enumerate(model._net.tables()):
        assert np.array_equal(w.detach().cpu().numpy().reshape(want[t].shape), want[t]), ('fit() vs engine replay', t)
    for t, w in enumerate(model._net.tables()):  # coarse drift sanity for the recordings without an engine-level replay
        ref = rec['final_%d' % t]
        ec.assert_open_loop_drift(w.detach().cpu().numpy().reshape(ref.shape), ref, (name, t))
    pred = model.predict(3)
    assert pred.dtype == np.float32 and pred.shape == (int(case['I']),)
    assert np.abs(pred - rec['predict_user3_all']).max() <= 2e-3 * np.abs(rec['predict_user3_all']).max()


def test_api_forms_pickle_resume():
    import io
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, 200, 5000).astype(np.int32), rs.randint(0, 300, 5000).astype(np.int32))
    model = ImplicitFactorizationModel(n_iter=2, batch_size=512, loss='bpr', optimizer_func=_adagrad,
                                       random_state=np.random.RandomState(1))
    model.fit(inter, verbose=True)
    items = np.arange(inter.num_items, dtype=np.int64)
    a, b = model.predict(1), model.predict(1, items)
    c = model.predict(np.repeat(1, inter.num_items).astype(np.int64), items)
    assert (a == b).all() and (b == c).all()
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    clone = torch.load(buf, weights_only=False)
    assert np.array_equal(clone.predict(2), model.predict(2))
    clone.fit(inter)  # resumes from pickled parameters + optimizer state


def test_training_learns_planted_structure():
    """End-to-end sanity at a size the oracle would take minutes for: users prefer items of
    their own cluster; after a few epochs the model must rank in-cluster items above
    out-of-cluster ones (a size-independent property, like the reference's MRR floors)."""
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    rs = np.random.RandomState(7)
    U, I, C, N = 20000, 5000, 10, 1_000_000
    users = rs.randint(0, U, N)
    items = (rs.randint(0, I // C, N) * C + users % C).astype(np.int32)  # item cluster == user cluster
    inter = Interactions(users.astype(np.int32), items, num_users=U, num_items=I)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=32, n_iter=4, batch_size=16384,
                                       optimizer_func=_adagrad, random_state=np.random.RandomState(3))
    model.fit(inter)
    wins = 0
    for u in range(0, 200):
        s = model.predict(u)
        own = s[np.arange(I) % C == u % C].mean()
        other = s[np.arange(I) % C != u % C].mean()
        wins += own > other
    assert wins >= 190, wins


@pytest.mark.parametrize('name', ['seq_bpr_adam_default', 'seq_hinge_adagrad_sparse', 'seq_pointwise_sparse_adam',
                                  'seq_adaptive_hinge_adagrad', 'seq_d64_bpr_adagrad', 'seq_d32_adaptive_adam',
                                  'seq_bloom_bpr_adagrad', 'seq_bloom_adaptive_hinge_adagrad', 'seq_bloom_d64_bpr_adagrad'])
def test_sequence_model_fit_predict_match_reference_run(name):
    """ImplicitSequenceModel (PoolNet) drop-in API on cuda:0 against the reference's recordings."""
    from test_host_seq_model import check_fit_predict_against_fixture
    model = check_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().cpu().numpy(), use_cuda=True)
    assert all(w.is_cuda for w in model._net.tables())


def test_sequence_model_learns_planted_structure():
    """Size-independent property at a size the oracle would not finish: sequences walk a ring of
    items (next = current + 1); after training, the model must rank the true next item of a
    held-out prefix far above a random item (cf. the reference's MRR floors,
    tests/sequence/test_sequence_implicit.py)."""
    from spotlight_amd.interactions import SequenceInteractions
    from spotlight_amd.sequence.implicit import ImplicitSequenceModel
    rs = np.random.RandomState(11)
    I, L, N = 500, 20, 20000
    start = rs.randint(1, I, N)
    seqs = ((start[:, None] + np.arange(L)[None, :] - 1) % (I - 1) + 1).astype(np.int32)
    model = ImplicitSequenceModel(loss='bpr', embedding_dim=32, n_iter=6, batch_size=512,
                                  optimizer_func=_adagrad, random_state=np.random.RandomState(3))
    model.fit(SequenceInteractions(seqs, num_items=I))
    wins = 0
    for k in range(100):
        prefix = seqs[k, :-1]
        scores = model.predict(prefix)
        nxt = seqs[k, -1]
        wins += (scores[nxt] > scores[1:]).mean() > 0.9  # top decile
    assert wins >= 80, wins


@pytest.mark.parametrize('name', ['bloom_item_bpr_adagrad', 'bloom_both_adaptive_adam',
                                  'bloom_user_pointwise_adagrad', 'bloom_c3_adaptive_adagrad'])
def test_bloom_model_fit_predict_match_reference_run(name):
    """BilinearNet with BloomEmbedding layers through the drop-in model API on cuda:0."""
    from test_host_model import check_bloom_fit_predict_against_fixture
    model = check_bloom_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().cpu().numpy(), use_cuda=True)
    assert all(w.is_cuda for w in model._net.tables())


@pytest.mark.parametrize('bloom', [False, True])
def test_ranking_metrics_fast_path_on_gpu(bloom):
    """evaluation.mrr_score / sequence_mrr_score through slk_bilinear_scores / slk_poolnet_scores /
    slk_rank_targets on cuda:0: bit-identical score rows, same MRRs as the per-user scipy route."""
    from test_host_api import check_factorization_fast_path, check_sequence_fast_path
    m = check_factorization_fast_path(bloom, use_cuda=True)
    assert all(w.is_cuda for w in m._net.tables())
    check_sequence_fast_path(bloom, use_cuda=True)


def test_mrr_fast_path_speed_at_movielens_100k_shape():
    """943 users x 1682 items, 20k held-out interactions (the shape of the reference's test split): the
    batched path must agree with the per-user route and is reported with its speed-up."""
    import time
    from spotlight_amd import evaluation as ev
    from spotlight_amd.cross_validation import random_train_test_split
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    from test_host_api import OnlyPredict
    rs = np.random.RandomState(42)
    inter = Interactions(rs.randint(0, 943, 100000).astype(np.int32), rs.randint(0, 1682, 100000).astype(np.int32),
                         num_users=943, num_items=1682)
    train, test = random_train_test_split(inter, random_state=np.random.RandomState(42))
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=32, n_iter=1, batch_size=1024, use_cuda=True,
                                       random_state=np.random.RandomState(1))
    model.fit(train)
    ev.mrr_score(model, test, train=train)  # warm-up
    t0 = time.perf_counter()
    fast = ev.mrr_score(model, test, train=train)
    t1 = time.perf_counter()
    slow = ev.mrr_score(OnlyPredict(model), test, train=train)
    t2 = time.perf_counter()
    assert np.allclose(fast, slow, rtol=1e-12, atol=0)
    print('mrr_score 943x1682: batched %.1f ms, per-user route %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))


def test_end_to_end_mrr_matches_reference_on_gpu():
    from test_host_api import check_end_to_end_mrr_matches_reference
    rec = np.load(os.path.join(GOLDEN, 'host_api.npz'))
    check_end_to_end_mrr_matches_reference(rec, use_cuda=True)


@pytest.mark.parametrize('name', ['explicit_regression_adam_default', 'explicit_poisson_adagrad',
                                  'explicit_logistic_sparse_adam', 'explicit_d64_regression_adagrad'])
def test_explicit_model_fit_predict_match_reference_run(name):
    """ExplicitFactorizationModel drop-in API on cuda:0 against the reference's recordings."""
    from test_host_explicit_model import check_fit_predict_against_fixture
    model = check_fit_predict_against_fixture(name, to_numpy=lambda w: w.detach().cpu().numpy(), use_cuda=True)
    assert all(w.is_cuda for w in model._net.tables())


def test_to_sequence_device_route_on_gpu():
    """Interactions.to_sequence(device='cuda') == the host route, then feeds the sequence model."""
    from spotlight_amd.interactions import Interactions
    from spotlight_amd.sequence.implicit import ImplicitSequenceModel
    rs = np.random.RandomState(9)
    n = 60000
    inter = Interactions(rs.randint(0, 900, n).astype(np.int32), rs.randint(1, 1500, n).astype(np.int32),
                         timestamps=rs.randint(0, 5000, n).astype(np.int32), num_users=900, num_items=1500)
    for kw in (dict(), dict(max_sequence_length=20, step_size=1), dict(max_sequence_length=7, min_sequence_length=3, step_size=2)):
        a, b = inter.to_sequence(**kw), inter.to_sequence(device='cuda', **kw)
        assert np.array_equal(a.sequences, b.sequences) and np.array_equal(a.user_ids, b.user_ids), kw
    seq = inter.to_sequence(max_sequence_length=12, min_sequence_length=4, step_size=3, device='cuda:0')
    model = ImplicitSequenceModel(loss='bpr', embedding_dim=16, n_iter=1, batch_size=256, use_cuda=True,
                                  random_state=np.random.RandomState(1))
    model.fit(seq)
    assert model.predict(seq.sequences[0]).shape == (1500,)


# ---- torch-side encoders fed by the embedding front-end (csrc/slk_embed.hip) ----
@pytest.mark.parametrize('name', ['enc_lstm_bpr_adam_default', 'enc_lstm_hinge_adagrad_sparse',
                                  'enc_lstm_adaptive_hinge_adagrad', 'enc_cnn_pointwise_adam_default',
                                  'enc_cnn_deep_bpr_adagrad', 'enc_mixture_bpr_adam_default',
                                  'enc_lstm_bloom_bpr_adagrad', 'enc_cnn_bloom_pointwise_adam_default',
                                  'enc_lstm_d32_bpr_adam_default'])
def test_encoder_models_match_reference_run_on_gpu(name):
    """LSTM / CNN / mixture representations on cuda:0 (encoder body on MIOpen, lookups + their backward on the
    package's kernels) against the reference's CPU recordings: identical initialisation and RandomState
    consumption, first-step gradients and losses within 1e-4, trained parameters within the trajectory band."""
    from test_host_encoders import check_against_fixture
    model = check_against_fixture(name, to_numpy=lambda t: t.detach().cpu().numpy(), tol=1e-4, traj_tol=5e-3, use_cuda=True)
    assert all(p.is_cuda for p in model._net.parameters())


@pytest.mark.parametrize('dim,sparse', [(1, False), (64, False), (64, True), (128, False), (20, True)])
def test_lookup_matches_torch_embedding_on_gpu(dim, sparse):
    """Front-end vs torch's own embedding autograd on the device at a training-sized batch (4096 x 50 lookups
    over 100 000 rows, Zipf-skewed so that some rows collect thousands of gradient rows)."""
    from spotlight_amd.embedding import lookup
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(dim)
    rows, shape = 100000, (4096, 50)
    w0 = torch.from_numpy(rs.normal(size=(rows, dim)).astype(np.float32)).to(dev)
    ids_np = np.minimum(rs.zipf(1.3, shape) - 1, rows - 1)
    ids = torch.from_numpy(ids_np).to(dev)
    upstream = torch.from_numpy(rs.normal(size=shape + (dim,)).astype(np.float32)).to(dev)
    w_ref = w0.clone().requires_grad_(True)
    out_ref = torch.nn.functional.embedding(ids, w_ref, padding_idx=0)
    (out_ref * upstream).sum().backward()
    w = w0.clone().requires_grad_(True)
    out = lookup(w, ids, padding_idx=0, sparse=sparse)
    assert torch.equal(out, out_ref)
    (out * upstream).sum().backward()
    g = w.grad.to_dense() if sparse else w.grad
    assert w.grad.is_sparse == sparse and float(g[0].abs().sum()) == 0.0
    # float64 reference of the same sums: both implementations must be within fp32 summation error of it
    exact = torch.zeros(rows, dim, dtype=torch.float64, device=dev)
    exact.index_add_(0, ids.reshape(-1), upstream.reshape(-1, dim).double())
    exact[0] = 0
    scale = exact.abs().max()
    assert float((g.double() - exact).abs().max() / scale) < 1e-5
    assert float((w_ref.grad.double() - exact).abs().max() / scale) < 1e-4


def test_pipelined_fit_is_value_neutral_on_gpu():
    """Epoch e + 1's shuffle and negatives drawn on a second ctx / HIP stream while epoch e trains: same tables, bit for bit."""
    from test_host_model import check_pipelined_fit_is_value_neutral
    check_pipelined_fit_is_value_neutral(use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())


def test_prefetched_first_chunk_is_value_neutral_on_gpu():
    """The next epoch's first chunk prepared on the ctx's second stream beside the last passes of this epoch
    (slk_bilinear_prefetch; real streams, real overlap): same tables and RandomState, bit for bit."""
    import torch
    from spotlight_amd.factorization import implicit as host
    from test_host_model import check_prefetched_first_chunk_is_value_neutral
    engine = host._engine_for(torch.device('cuda', 0))
    check_prefetched_first_chunk_is_value_neutral(engine, use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())


def test_bias_shadowed_fit_is_value_neutral_on_gpu():
    """fit() on the item-bias shadow (large item tables; forced here) against the plain layout: tables, accumulators and
    RandomState bit for bit."""
    import torch
    from spotlight_amd.factorization import implicit as host
    from test_host_model import check_bias_shadowed_fit_is_value_neutral
    engine = host._engine_for(torch.device('cuda', 0))
    check_bias_shadowed_fit_is_value_neutral(engine, use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())


def test_user_pingponged_fit_is_value_neutral_on_gpu():
    """fit() on the doubled user table (large minibatches; forced here) against the one-table layout: tables, optimizer state,
    predictions and RandomState bit for bit."""
    import torch
    from spotlight_amd.factorization import implicit as host
    from test_host_model import check_user_pingponged_fit_is_value_neutral
    engine = host._engine_for(torch.device('cuda', 0))
    check_user_pingponged_fit_is_value_neutral(engine, use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())


def test_pipelined_seq_fit_is_value_neutral_on_gpu():
    from test_host_seq_model import check_pipelined_seq_fit_is_value_neutral
    check_pipelined_seq_fit_is_value_neutral(use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())


def test_pipelined_explicit_fit_is_value_neutral_on_gpu():
    from test_host_explicit_model import check_pipelined_explicit_fit_is_value_neutral
    check_pipelined_explicit_fit_is_value_neutral(use_cuda=True, to_numpy=lambda w: w.detach().cpu().numpy())
