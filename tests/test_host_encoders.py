"""The torch-side encoders (LSTMNet / CNNNet / MixtureLSTMNet) behind ImplicitSequenceModel, fed by the
embedding front-end (spotlight_amd/embedding.py, csrc/slk_embed.hip), through the emulator build of the
kernels -- against the fixtures recorded from the live reference (oracle/make_golden_encoders.py): same seed =>
same initial parameters, same shuffles and negatives, same first-step gradients, losses and trained
parameters.  Plus the front-end on its own against torch's nn.functional.embedding autograd."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from emu_backend import emu_lib
from spotlight_amd import _native
from spotlight_amd.embedding import lookup
from spotlight_amd.factorization import implicit as host
from spotlight_amd.interactions import SequenceInteractions
from spotlight_amd.layers import BloomEmbedding, ScaledEmbedding, ZeroEmbedding
from spotlight_amd.sequence.implicit import ImplicitSequenceModel
from spotlight_amd.sequence.representations import CNNNet, LSTMNet, MixtureLSTMNet


@pytest.fixture()
def emu_device(monkeypatch):
    eng = _native.Engine(0, lib=emu_lib())
    monkeypatch.setattr(host, '_engine_for', lambda device: eng)
    monkeypatch.setattr(host, '_stream_for', lambda device: 0)
    monkeypatch.setattr(host, '_model_device', lambda: torch.device('cpu'))
    yield eng
    eng.close()


def _optimizer_func(opt):
    if opt == 'adam_default':
        return None
    return lambda params: torch.optim.Adagrad(params, lr=0.05)


def case_of(rec):
    case = {k[5:]: rec[k][()] for k in rec.files if k.startswith('case_')}
    case['rep_kw'] = {k[6:]: (tuple(rec[k].tolist()) if rec[k].ndim else rec[k][()].item() if hasattr(rec[k][()], 'item')
                              else rec[k][()]) for k in rec.files if k.startswith('repkw_')}
    return case


def build_model(case, **model_kw):
    """Constructed the way the fixture was: the constructor seeds torch, an explicit representation is built after it."""
    I, D = int(case['I']), int(case['D'])
    opt = str(case['opt'])
    model = ImplicitSequenceModel(
        loss=str(case['loss']), representation='pooling', embedding_dim=D, n_iter=int(case['n_iter']),
        batch_size=int(case['B']), l2=float(case.get('l2', 0.0)), learning_rate=float(case.get('lr', 1e-2)),
        optimizer_func=_optimizer_func(opt), sparse=opt == 'adagrad_sparse',
        random_state=np.random.RandomState(int(case['seed'])), num_negative_samples=int(case.get('n_neg', 5)), **model_kw)
    kind, kw = str(case['rep']), dict(case['rep_kw'])
    for key in ('nonlinearity',):
        if key in kw:
            kw[key] = str(kw[key])
    if 'bloom' in case:
        kw['item_embedding_layer'] = BloomEmbedding(I, D, compression_ratio=float(case['ratio']),
                                                    num_hash_functions=int(case['bloom']), padding_idx=0)
    if kw:
        model._representation = {'lstm': LSTMNet, 'cnn': CNNNet, 'mixture': MixtureLSTMNet}[kind](I, embedding_dim=D, **kw)
    else:
        model._representation = kind
    return model


def check_against_fixture(name, to_numpy=lambda t: t.detach().numpy(), tol=2e-5, traj_tol=2e-3, **model_kw):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_of(rec)
    model = build_model(case, **model_kw)
    inter = SequenceInteractions(rec['sequences'], num_items=int(case['I']))
    model._initialize(inter)
    params = dict(model._net.named_parameters())
    names = [str(n) for n in rec['names']]
    assert list(params) == names  # creation order and names of the reference
    for t, nm in enumerate(names):
        assert np.array_equal(to_numpy(params[nm]), rec['init_%d' % t]), nm

    first_grads, losses = [], []
    orig_step, orig_loss = model._optimizer.step, model._loss_func

    def rec_step(*a, **kw):
        if not first_grads:
            for nm in names:
                g = params[nm].grad
                first_grads.append(to_numpy(g.to_dense() if g.is_sparse else g).copy())
        return orig_step(*a, **kw)

    def rec_loss(*a, **kw):
        out = orig_loss(*a, **kw)
        losses.append(float(out.item()))
        return out

    model._optimizer.step, model._loss_func = rec_step, rec_loss
    model.fit(inter)
    st = model._random_state.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])  # same draws consumed
    assert len(losses) == len(rec['losses'])
    assert abs(losses[0] - rec['losses'][0]) <= tol * abs(rec['losses'][0])
    assert np.abs(np.array(losses) - rec['losses']).max() <= traj_tol * np.abs(rec['losses']).max()
    for t, nm in enumerate(names):
        want = rec['grad0_%d' % t]
        scale = max(np.abs(want).max(), 1e-12)
        assert np.abs(first_grads[t].reshape(want.shape) - want).max() <= tol * scale, ('first-step gradient', nm)
    for t, nm in enumerate(names):
        want = rec['final_%d' % t]
        bad = np.abs(to_numpy(params[nm]).reshape(want.shape) - want) > traj_tol * max(np.abs(want).max(), 1e-12)
        assert bad.mean() <= 0.05, ('trained parameters', nm, bad.mean())
    pred = model.predict(rec['predict_seq'])
    assert pred.dtype == np.float32 and pred.shape == rec['predict_all'].shape
    assert np.abs(pred - rec['predict_all']).max() <= 5 * traj_tol * max(np.abs(rec['predict_all']).max(), 1e-6)
    some = model.predict(rec['predict_seq2'], rec['predict_items'])
    assert some.shape == rec['predict_some'].shape
    assert np.abs(some - rec['predict_some']).max() <= 5 * traj_tol * max(np.abs(rec['predict_all']).max(), 1e-6)
    return model


@pytest.mark.parametrize('name', golden_names('enc'))
def test_encoder_models_match_reference_run(emu_device, name):
    check_against_fixture(name)


@pytest.mark.parametrize('dim', [1, 8, 20, 64, 3])
@pytest.mark.parametrize('sparse', [False, True])
def test_lookup_matches_torch_embedding(emu_device, dim, sparse):
    """Forward bit-identical to weight[ids]; backward == torch's embedding autograd (padding row excluded)."""
    rs = np.random.RandomState(dim)
    rows = 37
    w0 = torch.from_numpy(rs.normal(size=(rows, dim)).astype(np.float32))
    ids = torch.from_numpy(rs.randint(0, rows, (9, 13)))
    ids[0, :4] = 0
    upstream = torch.from_numpy(rs.normal(size=(9, 13, dim)).astype(np.float32))
    w_ref = w0.clone().requires_grad_(True)
    out_ref = torch.nn.functional.embedding(ids, w_ref, padding_idx=0)
    (out_ref * upstream).sum().backward()
    w = w0.clone().requires_grad_(True)
    out = lookup(w, ids, padding_idx=0, sparse=sparse)
    assert out.shape == out_ref.shape and torch.equal(out, out_ref)
    (out * upstream).sum().backward()
    g = w.grad
    if sparse:
        assert g.is_sparse
        assert torch.equal(g._indices()[0], torch.unique(ids[ids != 0]))  # distinct rows, ascending, no padding row
        g = g.to_dense()
    assert g[0].abs().sum() == 0
    assert torch.allclose(g, w_ref.grad, rtol=1e-6, atol=1e-6)


def test_layers_call_the_front_end(emu_device):
    rs = np.random.RandomState(0)
    torch.manual_seed(0)
    for layer in (ScaledEmbedding(30, 8, padding_idx=0), ZeroEmbedding(30, 1, padding_idx=0),
                  ScaledEmbedding(30, 8, padding_idx=0, sparse=True)):
        ids = torch.from_numpy(rs.randint(0, 30, (5, 6)))
        out = layer(ids)
        assert torch.equal(out, layer.weight[ids])
        out.sum().backward()
        assert layer.weight.grad.is_sparse == layer.sparse
    bloom = BloomEmbedding(50, 8, compression_ratio=0.4, num_hash_functions=3)
    ids = torch.from_numpy(rs.randint(0, 50, (4, 7)))
    ids[1, :3] = 0
    from sklearn.utils import murmurhash3_32
    flat = ids.reshape(-1).numpy().astype(np.int32)
    hashed = np.stack([murmurhash3_32(flat, seed=s) % bloom.compressed_num_embeddings for s in bloom._masks], 1)
    hashed[flat == 0] = 0
    hashed = torch.from_numpy(hashed.astype(np.int64))
    w_ref = bloom.weight.detach().clone().requires_grad_(True)
    want = torch.nn.functional.embedding(hashed, w_ref, padding_idx=0).sum(1).view(4, 7, 8)
    got = bloom(ids)
    assert got.shape == (4, 7, 8) and torch.allclose(got, want, rtol=0, atol=1e-7)
    assert bloom(ids[:, 0]).shape == (4, 1, 8)
    up = torch.from_numpy(rs.normal(size=(4, 7, 8)).astype(np.float32))
    (want * up).sum().backward()
    (got * up).sum().backward()
    assert torch.allclose(bloom.weight.grad, w_ref.grad, rtol=1e-6, atol=1e-6)


def test_front_end_argument_errors(emu_device):
    w = torch.zeros(10, 4)
    with pytest.raises(RuntimeError):
        lookup(w.double(), torch.zeros(3, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        lookup(torch.zeros(10, 8)[:, ::2], torch.zeros(3, dtype=torch.int64))
    eng = emu_device
    with pytest.raises(_native.SlkError):  # fill without a plan
        eng.embedding_backward_fill(w.data_ptr(), w.data_ptr())
    assert lookup(w, torch.zeros((0,), dtype=torch.int64)).shape == (0, 4)


@pytest.mark.parametrize('sparse', [False, True])
@pytest.mark.parametrize('n', [15, 16, 17, 33, 257, 6000])
def test_lookup_backward_under_skew(emu_device, n, sparse):
    """Runs much longer than a reduction chunk (one row owns 60 % of the lookups) go through several levels of
    the segmented reduction; sizes around the chunk edges."""
    rs = np.random.RandomState(n)
    rows, dim = 50, 8
    ids_np = rs.randint(0, rows, n)
    ids_np[rs.rand(n) < 0.6] = 7
    ids = torch.from_numpy(ids_np)
    up = torch.from_numpy(rs.normal(size=(n, dim)).astype(np.float32))
    w = torch.zeros(rows, dim, requires_grad=True)
    (lookup(w, ids, padding_idx=0, sparse=sparse) * up).sum().backward()
    g = w.grad.to_dense() if sparse else w.grad
    exact = torch.zeros(rows, dim, dtype=torch.float64)
    exact.index_add_(0, ids, up.double())
    exact[0] = 0
    assert float((g.double() - exact).abs().max()) <= 1e-5 * float(exact.abs().max())
    # bit-reproducible
    w2 = torch.zeros(rows, dim, requires_grad=True)
    (lookup(w2, ids, padding_idx=0, sparse=sparse) * up).sum().backward()
    assert torch.equal(w2.grad.to_dense() if sparse else w2.grad, g)
