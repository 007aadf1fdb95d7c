"""Parity of the engine against the CPU oracle AT THE BENCHMARKED CONFIGURATIONS (BASELINE.json configs[1..4];
VERDICT r01 "What's weak" 1): one minibatch of the real size on the real table shapes.

The oracle is a scalar restatement of the reference; tables of 10^7 (or 1.25 * 10^8) rows are out of its reach,
but a minibatch only touches the rows it names.  So the tables live on the device only, and the oracle runs on
the COMPACTED problem: the pre-step rows of the users / items the minibatch touches, gathered from the device,
with ids renumbered in ascending order.  Renumbering changes nothing the reference computes (summation order
follows minibatch position, not id), so the comparison is exact for every touched row; all other rows must come
back bit-identical.

Tolerances (BASELINE.json north_star: "bit-exact sampled negative indices under the same seed, fp32 loss/grad
within 1e-5 rel"):
  negatives, RNG state   bit-exact against numpy.random.RandomState.randint
  loss                   |d| <= 1e-5 * |loss|
  summed gradients       ||d||inf <= 1e-5 * ||g||inf per embedding table, bias tables against their joint norm
  updated parameters     per element: |d| <= 1e-5 * ||p||inf + lr * (1e-5 * ||g||inf) / (sqrt(state + g^2) + eps)
                         -- the second term is what the stated gradient tolerance turns into through Adagrad's
                         update p -= lr * g / (sqrt(sum) + eps); it is only large where the summed gradient is
                         ~0 and the accumulator is still 0 (first step: the update is lr * sign(g)), so the
                         ill-conditioned elements are identified by their gradient magnitude, not by a quota
  Adagrad state          |d| <= 1e-5 * ||s||inf + 2 * |g| * (1e-5 * ||g||inf)
  untouched rows         bit-identical
"""
import numpy as np
import torch

from oracle.oracle import BilinearOracle, BloomBilinearOracle, PoolNetOracle, bloom_desc
from spotlight_amd import _native

TOL = 1e-5


def _np(t):
    return t.detach().cpu().numpy()


def _rows(t, idx_dev):
    return _np(t.index_select(0, idx_dev))


def _assert_grad(got, want, scale, what):
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max() if got.size else 0.0
    assert err <= TOL * scale, (what, float(err), float(scale))


def _assert_adagrad_update(p_got, p_want, s_got, s_want, s_pre, g, g_scale, lr, eps, what):
    """See the module docstring: per-element bound derived from the gradient tolerance."""
    p_scale = max(float(np.abs(p_want).max()), 1e-30)
    dg = TOL * g_scale
    g64, s64 = g.astype(np.float64), s_pre.astype(np.float64)
    tol_p = TOL * p_scale + lr * dg / (np.sqrt(s64 + g64 * g64) + eps)
    bad = np.abs(p_got.astype(np.float64) - p_want.astype(np.float64)) > tol_p
    assert not bad.any(), (what, 'param', int(bad.sum()), float(np.abs(p_got - p_want).max()))
    s_scale = max(float(np.abs(s_want).max()), 1e-30)
    tol_s = TOL * s_scale + 2.0 * np.abs(g64) * dg
    bad = np.abs(s_got.astype(np.float64) - s_want.astype(np.float64)) > tol_s
    assert not bad.any(), (what, 'state', int(bad.sum()))
    # how many elements needed the conditioning term at all (reported, not asserted)
    return int((np.abs(p_got.astype(np.float64) - p_want.astype(np.float64)) > TOL * p_scale).sum())


def _assert_sparse_adam_update(p_got, p_want, m_got, m_want, v_got, v_want, m_pre, v_pre, g, g_scale, lr_t, b1, b2, eps, what):
    """SparseAdam (torch/optim/_functional.py:44-84): m' = m + (1-b1)(g-m); v' = v + (1-b2)(g^2-v);
    p -= lr_t * m' / (sqrt(v') + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t).  Per-element bound = what a gradient perturbation
    of dg = 1e-5 * ||g||inf moves the element by: d(update)/dg = lr_t * ((1-b1)/(sqrt(v')+eps) - m'(1-b2) g / (sqrt(v')(sqrt(v')+eps)^2));
    like Adagrad's it is only large where v' ~ 0, i.e. zero second moment and a ~0 gradient."""
    dg = TOL * g_scale
    g64, m64, v64 = g.astype(np.float64), m_pre.astype(np.float64), v_pre.astype(np.float64)
    m1 = m64 + (1 - b1) * (g64 - m64)
    v1 = np.maximum(v64 + (1 - b2) * (g64 * g64 - v64), 0.0)
    rt = np.sqrt(v1)
    sens = (1 - b1) / (rt + eps) + np.abs(m1) * (1 - b2) * np.abs(g64) / (np.maximum(rt, 1e-30) * (rt + eps) ** 2)
    p_scale = max(float(np.abs(p_want).max()), 1e-30)
    tol_p = TOL * p_scale + lr_t * dg * sens
    d = np.abs(p_got.astype(np.float64) - p_want.astype(np.float64))
    bad = d > tol_p
    assert not bad.any(), (what, 'param', int(bad.sum()), float(d.max()))
    tol_m = TOL * max(float(np.abs(m_want).max()), 1e-30) + (1 - b1) * dg
    bad = np.abs(m_got.astype(np.float64) - m_want.astype(np.float64)) > tol_m
    assert not bad.any(), (what, 'exp_avg', int(bad.sum()))
    tol_v = TOL * max(float(np.abs(v_want).max()), 1e-30) + (1 - b2) * 2.0 * np.abs(g64) * dg
    bad = np.abs(v_got.astype(np.float64) - v_want.astype(np.float64)) > tol_v
    assert not bad.any(), (what, 'exp_avg_sq', int(bad.sum()))
    return int((d > TOL * p_scale).sum())


def _changed_rows(after, before):
    d = after != before
    return torch.nonzero(d.any(dim=1) if d.dim() > 1 else d).squeeze(1)


def bilinear_minibatch_parity(engine, dev, stream, U, I, D, B, loss='bpr', nn=1, scale=None, trained=False,
                              bloom_rows=0, n_hash=4, seed=0, check_grads=True, tables=None, state=None,
                              users=None, items=None, rng_state=None, lr=1e-2, opt='adagrad', state2=None, step0=0,
                              bias_scale=0.1, pingpong=False, bias_shadow=False):
    """One minibatch of B interactions over U x I tables of dim D (item layer: BloomEmbedding with `bloom_rows`
    compressed rows when > 0), Adagrad(lr) or SparseAdam(lr) (`opt`; `step0` minibatches already taken).  Tables /
    optimizer state / ids / RNG state are generated here on the device unless given.  `pingpong` / `bias_shadow`: the real step
    runs inside a user-row ping-pong scope / an item-bias shadow (slk_user_pingpong_begin, slk_bias_shadow_begin: what fit() and
    bench.py open for minibatches / item tables this large), closed before anything is compared.  Returns a dict of diagnostics."""
    assert opt in ('adagrad', 'sparse_adam')
    betas = (0.9, 0.999)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + seed)
    rows_i = bloom_rows or I
    if tables is None:
        sc = scale if scale is not None else 1.0 / D
        tables = [torch.empty(U, D, device=dev).normal_(0, sc, generator=gen),
                  torch.empty(rows_i, D, device=dev).normal_(0, sc, generator=gen),
                  torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
        if trained:  # a model some way into training: non-trivial biases and accumulators
            tables[2].normal_(0, bias_scale, generator=gen)
            tables[3].normal_(0, bias_scale, generator=gen)
        if bloom_rows:
            tables[1][0] = 0  # the compressed table's padding row (layers.py:152-154)
    if state is None:
        state = [torch.zeros_like(t) for t in tables]
        if trained and opt == 'adagrad':
            for s in state:
                s.uniform_(0.01, 1.0, generator=gen)
        elif trained:  # first moments of either sign
            for s in state:
                s.normal_(0, 1e-3, generator=gen)
    if opt == 'sparse_adam' and state2 is None:
        state2 = [torch.zeros_like(t) for t in tables]
        if trained:
            for s in state2:
                s.uniform_(1e-8, 1e-5, generator=gen)
    if users is None:
        users = torch.randint(0, U, (B,), device=dev, dtype=torch.int64, generator=gen)
        items = torch.randint(0, I, (B,), device=dev, dtype=torch.int64, generator=gen)
    if rng_state is None:
        rng_state = np.random.RandomState(7 + seed).get_state()
    n_draw = B * (nn if loss == 'adaptive_hinge' else 1)
    rs = np.random.RandomState()
    rs.set_state(rng_state)
    want_neg = rs.randint(0, I, n_draw, dtype=np.int64)
    want_rng = rs.get_state()

    ib = _native.make_bloom(rows_i, n_hash) if bloom_rows else None
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D, item_bloom=ib)
    mb_loss = torch.zeros(1, device=dev)
    neg_out = torch.full((n_draw,), -1, device=dev, dtype=torch.int64)
    out = {}

    # ---- compaction (ids are needed on the host anyway)
    h_users, h_items = _np(users), _np(items)
    uu, uinv = np.unique(h_users, return_inverse=True)
    if bloom_rows:
        iu = None  # hashed rows depend on the item id itself: the (small) compressed table stays whole
        pos_c, neg_c = h_items, want_neg
    else:
        iu, iinv = np.unique(np.concatenate([h_items, want_neg]), return_inverse=True)
        pos_c, neg_c = iinv[:B], iinv[B:]
    d_uu = torch.from_numpy(uu).to(dev)
    d_iu = torch.from_numpy(iu).to(dev) if iu is not None else None
    sel = lambda t, idx: _rows(t, idx) if idx is not None else _np(t)
    pre_p = [sel(tables[0], d_uu), sel(tables[1], d_iu), sel(tables[2], d_uu), sel(tables[3], d_iu)]
    pre_s = [sel(state[0], d_uu), sel(state[1], d_iu), sel(state[2], d_uu), sel(state[3], d_iu)]
    pre_s2 = [sel(state2[0], d_uu), sel(state2[1], d_iu), sel(state2[2], d_uu), sel(state2[3], d_iu)] if state2 else None

    # ---- oracle on the compacted problem
    if bloom_rows:
        assert opt == 'adagrad'
        ora = BloomBilinearOracle(*pre_p, item_bloom=bloom_desc(n_hash), opt='adagrad', lr=lr)
        for t in range(4):
            ora.s1[t][...] = pre_s[t]
    else:
        ora = BilinearOracle(*pre_p, opt=opt, lr=lr, sparse_grads=True, state1=pre_s, state2=pre_s2, betas=betas, step=step0)
    want_loss, want_g = ora.step(uinv, pos_c, neg_c, loss=loss, n_neg=nn, want_grads=True)

    # ---- engine, gradients: ADAM_DENSE accumulate-only (lr = 0, beta1 = 0 => exp_avg == the summed gradient)
    if check_grads:
        m1 = [torch.zeros_like(t) for t in tables]
        m2 = [torch.zeros_like(t) for t in tables]
        op = _native.make_optim('adam_dense', [t.data_ptr() for t in m1], [t.data_ptr() for t in m2], lr=0.0,
                                betas=(0.0, 0.999))
        engine.rng_set_state(rng_state)
        engine.bilinear_train(tb, op, users.data_ptr(), items.data_ptr(), B, B, loss, nn, mb_loss.data_ptr(),
                              d_neg_out=neg_out.data_ptr(), stream=stream)
        assert np.array_equal(_np(neg_out), want_neg), 'negatives differ from numpy randint'
        got_loss = float(mb_loss.item())
        assert abs(got_loss - want_loss) <= TOL * abs(want_loss), (got_loss, want_loss)
        g_dev = [sel(m1[0], d_uu), sel(m1[1], d_iu), sel(m1[2], d_uu), sel(m1[3], d_iu)]
        _assert_grad(g_dev[0], want_g[0], np.abs(want_g[0]).max(), 'user embedding gradient')
        _assert_grad(g_dev[1], want_g[1], np.abs(want_g[1]).max(), 'item embedding gradient')
        bscale = max(np.abs(want_g[2]).max(), np.abs(want_g[3]).max())
        _assert_grad(g_dev[2], want_g[2], bscale, 'user bias gradient')
        _assert_grad(g_dev[3], want_g[3], bscale, 'item bias gradient')
        # no gradient outside the touched rows
        for t, idx in ((0, d_uu), (1, d_iu), (2, d_uu), (3, d_iu)):
            if idx is None:
                continue
            nz = m1[t] != 0
            nzr = torch.nonzero(nz.any(dim=1) if nz.dim() > 1 else nz).squeeze(1)
            assert bool(torch.isin(nzr, idx).all()), ('gradient outside the minibatch rows', t)
        del m1, m2, g_dev
        neg_out.fill_(-1)

    # ---- engine, the real step (Adagrad, as bench.py runs it; or SparseAdam)
    before = [t.clone() for t in tables] + [s.clone() for s in state] + ([s.clone() for s in state2] if state2 else [])
    op = _native.make_optim(opt, [s.data_ptr() for s in state], [s.data_ptr() for s in state2] if state2 else None, lr=lr,
                            betas=betas, step=step0)
    calls0 = engine.get_stat('pingpong_calls')
    with engine.bias_shadow(tb, op, stream=stream, enabled=bias_shadow):
        with engine.user_pingpong(tb, op, stream=stream, enabled=pingpong):
            engine.rng_set_state(rng_state)
            engine.bilinear_train(tb, op, users.data_ptr(), items.data_ptr(), B, B, loss, nn, mb_loss.data_ptr(),
                                  d_neg_out=neg_out.data_ptr(), stream=stream)
    assert engine.get_stat('pingpong_calls') - calls0 == (1 if pingpong else 0)
    assert np.array_equal(_np(neg_out), want_neg), 'negatives differ from numpy randint'
    got_rng = engine.rng_get_state()
    assert (got_rng[1] == want_rng[1]).all() and got_rng[2] == want_rng[2], 'RNG state after the draw'
    got_loss = float(mb_loss.item())
    assert abs(got_loss - want_loss) <= TOL * abs(want_loss), (got_loss, want_loss)
    assert op.step == step0 + 1
    out.update(loss=got_loss, loss_oracle=want_loss, users_touched=int(uu.size),
               items_touched=int(iu.size) if iu is not None else None)

    eps = 1e-10
    bscale = max(np.abs(want_g[2]).max(), np.abs(want_g[3]).max())
    gscale = [np.abs(want_g[0]).max(), np.abs(want_g[1]).max(), bscale, bscale]
    idxs = [d_uu, d_iu, d_uu, d_iu]
    cond = 0
    for t in range(4):
        p_got, s_got = sel(tables[t], idxs[t]), sel(state[t], idxs[t])
        shp = ora.p[t].shape
        if opt == 'adagrad':
            cond += _assert_adagrad_update(p_got.reshape(shp), ora.p[t], s_got.reshape(shp), ora.s1[t],
                                           pre_s[t].reshape(shp), want_g[t], gscale[t], lr, eps, ('table', t))
        else:
            tt = step0 + 1
            lr_t = lr * np.sqrt(1.0 - betas[1] ** tt) / (1.0 - betas[0] ** tt)
            v_got = sel(state2[t], idxs[t])
            cond += _assert_sparse_adam_update(p_got.reshape(shp), ora.p[t], s_got.reshape(shp), ora.s1[t], v_got.reshape(shp),
                                               ora.s2[t], pre_s[t].reshape(shp), pre_s2[t].reshape(shp), want_g[t], gscale[t],
                                               lr_t, betas[0], betas[1], 1e-8, ('table', t))
        # rows the minibatch does not name: bit-identical
        pairs = [(tables[t], before[t], 'param'), (state[t], before[4 + t], 'state')]
        if state2:
            pairs.append((state2[t], before[8 + t], 'state2'))
        for after, bef, nm in pairs:
            ch = _changed_rows(after, bef)
            if idxs[t] is not None:
                assert bool(torch.isin(ch, idxs[t]).all()), ('row outside the minibatch changed', nm, t)
    out['elements_beyond_1e-5_but_within_conditioned_bound'] = cond
    return out


def saturated_problem(dev, U, I, D, B, seed=0, row_scale=0.35, shift=2.0):
    """A model that has LEARNED something, for the minibatch parity check (the N(0, 1/16) 'trained' state
    keeps every score inside the sigmoid's linear regime, loss 0.49998): rows N(0, row_scale^2) -- a dim-64 dot product has a
    standard deviation of 8 * row_scale^2 ~ 1 --, item biases +shift on the lower half of the ids and -shift on the upper half
    (+ noise), positives drawn from the lower half only, negatives uniform as the reference draws them.  pos - neg is then
    ~N(0, 1.5^2) for half of the pairs and ~N(2 * shift, 1.5^2) for the other half: |pos - neg| spans 0 .. 8, s(1 - s) goes
    down to 1e-4 (saturated pairs, where the cancellation in the summed gradients is hardest) and the bpr loss is ~0.27."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(7000 + seed)
    tables = [torch.empty(U, D, device=dev).normal_(0, row_scale, generator=gen),
              torch.empty(I, D, device=dev).normal_(0, row_scale, generator=gen),
              torch.empty(U, device=dev).normal_(0, 0.3, generator=gen),
              torch.empty(I, device=dev).normal_(0, 0.3, generator=gen)]
    half = I // 2
    tables[3][:half] += shift
    tables[3][half:] -= shift
    state = [torch.empty_like(t).uniform_(0.01, 1.0, generator=gen) for t in tables]
    users = torch.randint(0, U, (B,), device=dev, dtype=torch.int64, generator=gen)
    items = torch.randint(0, max(half, 1), (B,), device=dev, dtype=torch.int64, generator=gen)
    return tables, state, users, items


def poolnet_minibatch_parity(engine, dev, stream, I, D, B, L, loss='bpr', nn=1, scale=None, trained=False, seed=0,
                             pad_frac=0.0, lr=1e-2, check_grads=True):
    """One minibatch of B sequences of length L (C4: 4096 x 200 over 10^6 items, dim 64), Adagrad(lr): PoolNet's
    two tables are small enough (256 MB) for the oracle to hold them whole."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(2000 + seed)
    sc = scale if scale is not None else 1.0 / D
    E = torch.empty(I, D, device=dev).normal_(0, sc, generator=gen)
    bias = torch.zeros(I, device=dev)
    if trained:
        bias.normal_(0, 0.1, generator=gen)
    E[0] = 0
    bias[0] = 0
    st = [torch.zeros_like(E), torch.zeros_like(bias)]
    if trained:
        for s in st:
            s.uniform_(0.01, 1.0, generator=gen)
    seqs = torch.randint(1, I, (B, L), device=dev, dtype=torch.int64, generator=gen)
    if pad_frac > 0:
        npad = torch.randint(0, L, (B,), device=dev, generator=gen)
        npad = torch.where(torch.rand(B, device=dev, generator=gen) < pad_frac, npad, torch.zeros_like(npad))
        seqs[torch.arange(L, device=dev)[None, :] < npad[:, None]] = 0  # left padding
    rng_state = np.random.RandomState(11 + seed).get_state()
    n_draw = B * L * (nn if loss == 'adaptive_hinge' else 1)
    rs = np.random.RandomState()
    rs.set_state(rng_state)
    want_neg = rs.randint(0, I, n_draw, dtype=np.int64)
    want_rng = rs.get_state()

    pre_p, pre_s = [_np(E), _np(bias)], [_np(st[0]), _np(st[1])]
    ora = PoolNetOracle(pre_p[0], pre_p[1], opt='adagrad', lr=lr, state1=pre_s)
    h_seqs = _np(seqs)
    want_loss, want_g = ora.step(h_seqs, want_neg, loss=loss, n_neg=nn, want_grads=True)

    tb = _native.make_seq_tables(E.data_ptr(), bias.data_ptr(), I, D)
    mb_loss = torch.zeros(1, device=dev)
    neg_out = torch.full((n_draw,), -1, device=dev, dtype=torch.int64)
    if check_grads:
        m1, m2 = [torch.zeros_like(E), torch.zeros_like(bias)], [torch.zeros_like(E), torch.zeros_like(bias)]
        op = _native.make_optim('adam_dense', [None, m1[0].data_ptr(), None, m1[1].data_ptr()],
                                [None, m2[0].data_ptr(), None, m2[1].data_ptr()], lr=0.0, betas=(0.0, 0.999))
        engine.rng_set_state(rng_state)
        engine.poolnet_train(tb, op, 0, seqs.data_ptr(), B, L, B, loss, nn, mb_loss.data_ptr(),
                             d_neg_out=neg_out.data_ptr(), stream=stream)
        assert np.array_equal(_np(neg_out), want_neg)
        got_loss = float(mb_loss.item())
        assert abs(got_loss - want_loss) <= TOL * abs(want_loss), (got_loss, want_loss)
        _assert_grad(_np(m1[0]), want_g[0], np.abs(want_g[0]).max(), 'item embedding gradient')
        _assert_grad(_np(m1[1]), want_g[1], np.abs(want_g[1]).max(), 'item bias gradient')
        del m1, m2
        neg_out.fill_(-1)
    op = _native.make_optim('adagrad', [None, st[0].data_ptr(), None, st[1].data_ptr()], None, lr=lr)
    engine.rng_set_state(rng_state)
    engine.poolnet_train(tb, op, 0, seqs.data_ptr(), B, L, B, loss, nn, mb_loss.data_ptr(),
                         d_neg_out=neg_out.data_ptr(), stream=stream)
    assert np.array_equal(_np(neg_out), want_neg)
    got_rng = engine.rng_get_state()
    assert (got_rng[1] == want_rng[1]).all() and got_rng[2] == want_rng[2]
    got_loss = float(mb_loss.item())
    assert abs(got_loss - want_loss) <= TOL * abs(want_loss), (got_loss, want_loss)
    cond = 0
    for t, (p, s) in enumerate(((E, st[0]), (bias, st[1]))):
        cond += _assert_adagrad_update(_np(p).reshape(ora.p[t].shape), ora.p[t], _np(s).reshape(ora.s1[t].shape), ora.s1[t],
                                       pre_s[t].reshape(ora.s1[t].shape), want_g[t], np.abs(want_g[t]).max(), lr, 1e-10,
                                       ('seq table', t))
    assert bool((E[0] == 0).all()) and float(bias[0]) == 0.0  # padding row never trained
    return dict(loss=got_loss, loss_oracle=want_loss, timesteps=int((h_seqs != 0).sum()),
                **{'elements_beyond_1e-5_but_within_conditioned_bound': cond})


def multi_chunk_parity(engine, dev, stream, U, I, D, B, n_full, tail, check_at, seed=0, lr=1e-2, expect_route=None, loss='bpr', n_neg=1,
                       pingpong=False):
    """One slk_bilinear_train call over n_full * B + tail interactions (more than one prep chunk), bpr + Adagrad:
      * every negative of the call and the RNG state afterwards: bit-exact against numpy (one contiguous stream);
      * for every k in `check_at`: minibatch k is checked against the oracle by teacher forcing -- a second run
        over the first k minibatches only gives the engine's tables before minibatch k; the oracle takes ONE step
        from those (compacted) rows and must land on the full run's tables after minibatch k ... which are the
        tables of a run over k + 1 minibatches (the engine is deterministic and minibatch m's arithmetic does not
        depend on how many minibatches follow it in the call: also asserted, bit for bit, on the losses).
    """
    gen = torch.Generator(device=dev)
    gen.manual_seed(3000 + seed)
    N = n_full * B + tail
    T0 = [torch.empty(U, D, device=dev).normal_(0, 0.5 / np.sqrt(D), generator=gen),
          torch.empty(I, D, device=dev).normal_(0, 0.5 / np.sqrt(D), generator=gen),
          torch.empty(U, device=dev).normal_(0, 0.1, generator=gen), torch.empty(I, device=dev).normal_(0, 0.1, generator=gen)]
    S0 = [torch.empty_like(t).uniform_(0.01, 1.0, generator=gen) for t in T0]
    users = torch.randint(0, U, (N,), device=dev, dtype=torch.int64, generator=gen)
    items = torch.randint(0, I, (N,), device=dev, dtype=torch.int64, generator=gen)
    rng0 = np.random.RandomState(21 + seed).get_state()
    rs = np.random.RandomState()
    rs.set_state(rng0)
    n_mb = n_full + (1 if tail else 0)
    sizes = [B] * n_full + ([tail] if tail else [])
    nn = n_neg if loss == 'adaptive_hinge' else 1  # adaptive hinge: n draws per interaction (implicit.py:266-275)
    want_neg = np.concatenate([rs.randint(0, I, m * nn, dtype=np.int64) for m in sizes])  # one randint per minibatch
    want_rng = rs.get_state()

    def run(n_inter):
        t = [x.clone() for x in T0]
        s = [x.clone() for x in S0]
        tb = _native.make_tables([x.data_ptr() for x in t], U, I, D)
        op = _native.make_optim('adagrad', [x.data_ptr() for x in s], None, lr=lr)
        k = (n_inter + B - 1) // B
        mb = torch.zeros(k, device=dev)
        neg = torch.full((n_inter * nn,), -1, device=dev, dtype=torch.int64)
        with engine.user_pingpong(tb, op, stream=stream, enabled=pingpong):  # (the whole call inside a user-row ping-pong scope)
            engine.rng_set_state(rng0)
            engine.bilinear_train(tb, op, users.data_ptr(), items.data_ptr(), n_inter, B, loss, nn, mb.data_ptr(),
                                  d_neg_out=neg.data_ptr(), stream=stream)
        return t, s, _np(mb), neg, engine.rng_get_state(), op.step

    if expect_route is not None:  # 'epoch' (the persistent kernel) / 'launch': the route the call takes is part of what is tested
        engine.profile_reset()
        engine.profile_enable(True)
    tF, sF, lossF, negF, rngF, steps = run(N)
    if expect_route is not None:
        engine.profile_enable(False)
        prof = engine.profile_read()
        took_epoch = prof['epoch'][0] > 0
        assert took_epoch == (expect_route == 'epoch'), (expect_route, {k: v[0] for k, v in prof.items()})
    assert steps == n_mb
    assert np.array_equal(_np(negF), want_neg), 'negatives of the multi-chunk call differ from numpy'
    assert (rngF[1] == want_rng[1]).all() and rngF[2] == want_rng[2]
    assert np.isfinite(lossF).all()
    out = {'minibatches': n_mb, 'interactions': N, 'checked': []}
    for k in check_at:
        lo = k * B
        hi = min(lo + B, N)
        tA, sA, lossA, _, _, _ = run(lo)          # tables before minibatch k
        tB, sB, lossB, _, _, _ = run(hi)          # ... and after it
        assert np.array_equal(lossA, lossF[:k]) and np.array_equal(lossB, lossF[:k + 1]), 'a prefix run is not bit-identical'
        h_u, h_i, h_n = _np(users[lo:hi]), _np(items[lo:hi]), want_neg[lo * nn:hi * nn]
        uu, uinv = np.unique(h_u, return_inverse=True)
        iu, iinv = np.unique(np.concatenate([h_i, h_n]), return_inverse=True)
        d_uu, d_iu = torch.from_numpy(uu).to(dev), torch.from_numpy(iu).to(dev)
        idxs = [d_uu, d_iu, d_uu, d_iu]
        pre_p = [_rows(tA[t], idxs[t]) for t in range(4)]
        pre_s = [_rows(sA[t], idxs[t]) for t in range(4)]
        ora = BilinearOracle(*pre_p, opt='adagrad', lr=lr, sparse_grads=True, state1=pre_s)
        want_loss, want_g = ora.step(uinv, iinv[:hi - lo], iinv[hi - lo:], loss=loss, n_neg=nn, want_grads=True)
        assert abs(float(lossF[k]) - want_loss) <= TOL * abs(want_loss), (k, float(lossF[k]), want_loss)
        bscale = max(np.abs(want_g[2]).max(), np.abs(want_g[3]).max())
        gscale = [np.abs(want_g[0]).max(), np.abs(want_g[1]).max(), bscale, bscale]
        for t in range(4):
            _assert_adagrad_update(_rows(tB[t], idxs[t]).reshape(ora.p[t].shape), ora.p[t],
                                   _rows(sB[t], idxs[t]).reshape(ora.s1[t].shape), ora.s1[t],
                                   pre_s[t].reshape(ora.s1[t].shape), want_g[t], gscale[t], lr, 1e-10, ('minibatch', k, 'table', t))
            assert bool(torch.isin(_changed_rows(tB[t], tA[t]), idxs[t]).all())
        if hi == N:  # the full run ends here: its tables are the (k+1)-minibatch run's, bit for bit
            for t in range(4):
                assert torch.equal(tB[t], tF[t]) and torch.equal(sB[t], sF[t])
        out['checked'].append({'minibatch': k, 'loss': float(lossF[k]), 'loss_oracle': want_loss})
        del tA, sA, tB, sB
    return out


def sharded_world1_vs_fused(engine, dev, stream, U, I, D, B, n_mb=1, seed=0, lr=1e-2, block_rows=1 << 23):
    """The row-sharded exchange path at world 1 (slk_shard_* + ShardedBilinearTrainer; every exchange is a local copy)
    against the fused path (slk_bilinear_train) on the same tables, ids, RNG state: `n_mb` minibatches of B, bpr +
    Adagrad from a trained state (accumulators >= 0.01, so a 1e-5-relative gradient difference moves no element by more
    than 1e-6 of the table's norm).  Both paths sum a row's contributions before ONE update, in different association, so
    the comparison is |d| <= 1e-5 * ||p||inf per element, row by row over the WHOLE tables (the fused path's own parity
    with the oracle, including untouched rows staying bit-identical, is bilinear_minibatch_parity's job).  Needs an
    initialised torch.distributed group of world size 1.  Tables are generated twice from one seed rather than cloned
    (C5 shard: 32 GB item table + 32 GB accumulators per copy)."""
    import torch.distributed as dist
    from spotlight_amd.factorization.sharded import ShardedBilinearTrainer
    assert dist.is_initialized() and dist.get_world_size() == 1

    def fresh():
        gen = torch.Generator(device=dev)
        gen.manual_seed(9000 + seed)
        t = [torch.empty(U, D, device=dev).normal_(0, 0.5 / np.sqrt(D), generator=gen),
             torch.empty(I, D, device=dev).normal_(0, 0.5 / np.sqrt(D), generator=gen),
             torch.empty(U, device=dev).normal_(0, 0.1, generator=gen), torch.empty(I, device=dev).normal_(0, 0.1, generator=gen)]
        s = [torch.empty_like(x).uniform_(0.01, 1.0, generator=gen) for x in t]
        users = torch.randint(0, U, (n_mb * B,), device=dev, dtype=torch.int64, generator=gen)
        items = torch.randint(0, I, (n_mb * B,), device=dev, dtype=torch.int64, generator=gen)
        return t, s, users, items

    rng0 = np.random.RandomState(31 + seed).get_state()
    tF, sF, users, items = fresh()
    tb = _native.make_tables([x.data_ptr() for x in tF], U, I, D)
    opF = _native.make_optim('adagrad', [x.data_ptr() for x in sF], None, lr=lr)
    lossF = torch.zeros(n_mb, device=dev)
    engine.rng_set_state(rng0)
    engine.bilinear_train(tb, opF, users.data_ptr(), items.data_ptr(), n_mb * B, B, 'bpr', 1, lossF.data_ptr(), stream=stream)
    rngF = engine.rng_get_state()

    tS, sS, users2, items2 = fresh()
    assert torch.equal(users, users2) and torch.equal(items, items2)
    opS = _native.make_optim('adagrad', [x.data_ptr() for x in sS], None, lr=lr)
    tr = ShardedBilinearTrainer(engine, tS, opS, I, stream=stream)
    tr.reserve(B, n_mb)
    lossS = torch.zeros(n_mb, device=dev)
    engine.rng_set_state(rng0)
    tr.train(users, items, B, loss='bpr', mb_loss=lossS, sample_chunk=n_mb)
    rngS = engine.rng_get_state()
    assert (rngF[1] == rngS[1]).all() and rngF[2] == rngS[2], 'the two paths consumed the RNG stream differently'
    lf, ls = _np(lossF).astype(np.float64), _np(lossS).astype(np.float64)
    assert np.all(np.abs(lf - ls) <= TOL * np.abs(lf)), (lf, ls)
    assert opF.step == opS.step == n_mb
    out = {'loss_fused': lf.tolist(), 'loss_sharded_world1': ls.tolist(), 'max_abs_diff': {}, 'rows_differing': {}}
    for nm, a_list, b_list in (('param', tF, tS), ('adagrad_state', sF, sS)):
        for t in range(4):
            a, b = a_list[t], b_list[t]
            scale = float(a.abs().max())
            worst, nrows = 0.0, 0
            for r0 in range(0, a.shape[0], block_rows):
                d = (a[r0:r0 + block_rows] - b[r0:r0 + block_rows]).abs()
                worst = max(worst, float(d.max()))
                nz = d != 0
                nrows += int((nz.any(dim=1) if nz.dim() > 1 else nz).sum())
                del d, nz
            assert worst <= TOL * scale, (nm, t, worst, scale)
            out['max_abs_diff']['%s%d' % (nm, t)] = worst
            out['rows_differing']['%s%d' % (nm, t)] = nrows
    return out
