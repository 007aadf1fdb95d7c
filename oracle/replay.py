"""Replay of a recorded reference run (tests/golden/*.npz) through the C oracle.

TEST INFRASTRUCTURE (see oracle/slk_oracle.c).  Shared by oracle/make_golden.py (which
records the runs from the live reference) and tests/test_oracle.py (which re-checks the
committed fixtures where /root/reference does not exist).
"""
import numpy as np

from oracle.oracle import BilinearOracle, Rng

ORACLE_OPT = {'adam_default': 'adam_dense', 'adagrad': 'adagrad', 'adagrad_sparse': 'adagrad',
              'sparse_adam': 'sparse_adam', 'adagrad_dense_wd': 'adagrad_dense', 'sgd': 'sgd', 'sgd_sparse': 'sgd'}


def oracle_hparams(case):
    hp = _oracle_hparams(case)
    hp['sparse_grads'] = case['opt'] in ('adagrad_sparse', 'sparse_adam', 'sgd_sparse')
    return hp


def _oracle_hparams(case):
    opt = case['opt']
    if opt == 'adam_default':
        return dict(lr=case.get('lr', 1e-2), weight_decay=case.get('l2', 0.0))
    if opt == 'adagrad_dense_wd':
        return dict(lr=0.05, weight_decay=1e-3)
    if opt == 'sparse_adam':
        return dict(lr=0.01)
    return dict(lr=0.05)


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def frac_outside(a, b, tol=2e-4):
    """Open-loop DRIFT of a table after a multi-epoch replay: ||a - b||_2 / ||b||_2.  (Round 4: the former "fraction of the
    elements outside 2e-4" -- an outlier quota -- is gone; several epochs from zero accumulators are chaotic element by
    element, torch's own dense and sparse paths end 6e-4 apart, so the quota-free statement is a norm: a handful of
    sign-flipped lr-sized steps are invisible in it, a wrong update rule or a missed row is not.  The element-wise statements
    are the first-step gradients at 1e-5 here and the closed-loop engine tests, tests/engine_checks.py.)"""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def replay_with_oracle(case, rec, tol=1e-5):
    """Replays the recorded run through the C oracle; returns max parity errors."""
    o = BilinearOracle(rec['init_0'], rec['init_1'], rec['init_2'], rec['init_3'],
                       opt=ORACLE_OPT[case['opt']], **oracle_hparams(case))
    rng = Rng(state=('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    nn = case.get('n_neg', 5) if case['loss'] == 'adaptive_hinge' else 1
    losses, negs = [], []
    users64, items64 = rec['users'].astype(np.int64), rec['items'].astype(np.int64)
    errs = {}
    for e in range(case['n_iter']):
        perm = rng.shuffle_perm(case['N'])
        su, si = users64[perm], items64[perm]
        assert (su == rec['shuffled_users'][e]).all() and (si == rec['shuffled_items'][e]).all()
        if e == 0:
            # first minibatch gradients via a throw-away copy
            o2 = BilinearOracle(rec['init_0'], rec['init_1'], rec['init_2'], rec['init_3'],
                                opt=ORACLE_OPT[case['opt']], **oracle_hparams(case))
            B0 = min(case['B'], case['N'])
            _, dg = o2.step(su[:B0], si[:B0], rec['negatives'][:B0 * nn], loss=case['loss'],
                            n_neg=nn, want_grads=True)
            # bias-gradient scale: user-bias gradients are sums of +g/-g terms that cancel
            # (exactly for bpr/hinge), so both bias tables are judged against their joint norm
            bscale = max(np.abs(rec['grad0_2']).max(), np.abs(rec['grad0_3']).max())
            for t in range(4):
                ref = rec['grad0_%d' % t]
                if t < 2:
                    errs['grad0_%d' % t] = rel_inf(dg[t].reshape(ref.shape), ref)
                else:
                    errs['grad0_%d' % t] = np.abs(dg[t].reshape(ref.shape) - ref).max() / bscale
        l, ng = o.train(rng, su, si, case['B'], loss=case['loss'], n_neg=nn, want_negs=True)
        losses.append(l)
        negs.append(ng)
    negs = np.concatenate(negs)
    assert (negs == rec['negatives']).all(), 'negative ids differ'
    errs['loss'] = np.max(np.abs(np.concatenate(losses) - rec['losses']) / np.abs(rec['losses']))
    # the two bias tables are judged against their JOINT norm, like their gradients: the user-bias gradient is a sum of
    # +g / -g terms that cancel (exactly for bpr / hinge; to ~1e-3 of a term for the pointwise loss while the scores are
    # small), so with a linear optimizer (SGD) the user biases themselves stay at the level of that residue -- 1e-9 next to
    # item biases of 1e-3 -- and a comparison relative to their own norm would be one of rounding noise with rounding noise
    def rel_bias(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = max(np.abs(rec['final_2']).max(), np.abs(rec['final_3']).max(), 1e-30)
        return np.abs(a - b).max() / scale
    for t in range(4):
        if t >= 2:
            errs['final_%d' % t] = rel_bias(o.p[t].reshape(rec['final_%d' % t].shape), rec['final_%d' % t])
            continue
        errs['final_%d' % t] = rel_inf(o.p[t].reshape(rec['final_%d' % t].shape), rec['final_%d' % t])
    for t in range(4):
        errs['state1_%d' % t] = rel_inf(o.s1[t].reshape(rec['state1_%d' % t].shape), rec['state1_%d' % t])
        if 'state2_%d' % t in rec:
            errs['state2_%d' % t] = rel_inf(o.s2[t].reshape(rec['state2_%d' % t].shape),
                                            rec['state2_%d' % t])
    st = rng.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    fr = {}
    for t in range(4):
        ref = rec['final_%d' % t]
        got = o.p[t].reshape(ref.shape)
        if t >= 2:
            # (both bias tables against their JOINT norm, like their gradients above)
            bnorm = max(np.sqrt(np.linalg.norm(np.asarray(rec['final_2'], np.float64)) ** 2 + np.linalg.norm(np.asarray(rec['final_3'], np.float64)) ** 2), 1e-30)
            fr['final_%d' % t] = float(np.linalg.norm(np.asarray(got, np.float64).ravel() - np.asarray(ref, np.float64).ravel()) / bnorm)
        else:
            fr['final_%d' % t] = frac_outside(got, ref)
    errs['predict_all'] = rel_inf(o.predict(3), rec['predict_user3_all'])
    errs['predict_pairs'] = rel_inf(o.predict(rec['predict_pairs_u'], rec['predict_pairs_i']),
                                    rec['predict_pairs'])
    return errs, fr


def case_from_rec(rec):
    """Rebuilds the case dict stored alongside a fixture."""
    case = {}
    for k in rec.files if hasattr(rec, 'files') else rec.keys():
        if k.startswith('case_'):
            v = rec[k]
            case[k[5:]] = v.item() if v.shape == () else v
    return case


def replay_seq_with_oracle(case, rec):
    """Replays a recorded ImplicitSequenceModel(PoolNet) run through the C oracle."""
    from oracle.oracle import PoolNetOracle, bloom_desc
    hp = _oracle_hparams(case)
    bloom = bloom_desc(int(case['bloom'])) if ('bloom' in case and int(case['bloom'])) else None
    mk = lambda: PoolNetOracle(rec['init_0'], rec['init_1'], opt=ORACLE_OPT[case['opt']], item_bloom=bloom, **hp)
    o = mk()
    rng = Rng(state=('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    nn = int(case.get('n_neg', 5)) if case['loss'] == 'adaptive_hinge' else 1
    N, L, B = int(case['N']), int(case['L']), int(case['B'])
    seqs64 = rec['sequences'].astype(np.int64)
    losses, negs, errs = [], [], {}
    for e in range(int(case['n_iter'])):
        # sequence/implicit.py:215-216 rebinds `sequences` to its shuffled copy, so the
        # permutations of successive epochs COMPOSE (unlike the factorization model's)
        perm = rng.shuffle_perm(N)
        seqs64 = seqs64[perm]
        ss = seqs64
        assert (ss == rec['shuffled'][e]).all()
        if e == 0:
            B0 = min(B, N)
            l0, dg = mk().step(ss[:B0], rec['negatives'][:B0 * nn * L], loss=str(case['loss']), n_neg=nn,
                               want_grads=True)
            errs['loss0'] = abs(l0 - rec['losses'][0]) / abs(rec['losses'][0])
            for t in range(2):
                ref = rec['grad0_%d' % t]
                errs['grad0_%d' % t] = rel_inf(dg[t].reshape(ref.shape), ref)
        l, ng = o.train(rng, ss, B, loss=str(case['loss']), n_neg=nn, want_negs=True)
        losses.append(l)
        negs.append(ng)
    assert (np.concatenate(negs) == rec['negatives']).all(), 'negative ids differ'
    errs['loss'] = np.max(np.abs(np.concatenate(losses) - rec['losses']) / np.abs(rec['losses']))
    fr = {}
    for t in range(2):
        ref = rec['final_%d' % t]
        errs['final_%d' % t] = rel_inf(o.p[t].reshape(ref.shape), ref)
        fr['final_%d' % t] = frac_outside(o.p[t].reshape(ref.shape), ref)
        errs['state1_%d' % t] = rel_inf(o.s1[t].reshape(ref.shape), rec['state1_%d' % t])
    st = rng.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    # predictions from the reference's own final tables (the trajectory may legitimately drift)
    po = PoolNetOracle(rec['final_0'], rec['final_1'], item_bloom=bloom)
    errs['predict_all'] = rel_inf(po.predict(rec['predict_seq']), rec['predict_all'])
    errs['predict_some'] = rel_inf(po.predict(rec['predict_seq2'], rec['predict_items']), rec['predict_some'])
    return errs, fr


def bloom_oracle_for(case, rec, which='init'):
    from oracle.oracle import BloomBilinearOracle, bloom_desc
    mk = lambda on: bloom_desc(n_hash=int(case['H']), bag=bool(case.get('bag', False))) if on else None
    return BloomBilinearOracle(rec[which + '_0'], rec[which + '_1'], rec[which + '_2'], rec[which + '_3'],
                               user_bloom=mk(int(case['user_bloom'])), item_bloom=mk(int(case['item_bloom'])),
                               opt=ORACLE_OPT[str(case['opt'])], **_oracle_hparams(case))


def replay_bloom_with_oracle(case, rec):
    """Replays a recorded BloomEmbedding BilinearNet run through the C oracle."""
    o = bloom_oracle_for(case, rec)
    rng = Rng(state=('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    nn = int(case.get('n_neg', 5)) if case['loss'] == 'adaptive_hinge' else 1
    N, B = int(case['N']), int(case['B'])
    users64, items64 = rec['users'].astype(np.int64), rec['items'].astype(np.int64)
    losses, negs, errs = [], [], {}
    for e in range(int(case['n_iter'])):
        perm = rng.shuffle_perm(N)
        su, si = users64[perm], items64[perm]
        assert (su == rec['shuffled_users'][e]).all() and (si == rec['shuffled_items'][e]).all()
        if e == 0:
            B0 = min(B, N)
            l0, dg = bloom_oracle_for(case, rec).step(su[:B0], si[:B0], rec['negatives'][:B0 * nn],
                                                      loss=str(case['loss']), n_neg=nn, want_grads=True)
            errs['loss0'] = abs(l0 - rec['losses'][0]) / abs(rec['losses'][0])
            bscale = max(np.abs(rec['grad0_2']).max(), np.abs(rec['grad0_3']).max())
            for t in range(4):
                ref = rec['grad0_%d' % t]
                errs['grad0_%d' % t] = (rel_inf(dg[t].reshape(ref.shape), ref) if t < 2
                                        else np.abs(dg[t].reshape(ref.shape) - ref).max() / bscale)
        l, ng = o.train(rng, su, si, B, loss=str(case['loss']), n_neg=nn, want_negs=True)
        losses.append(l)
        negs.append(ng)
    assert (np.concatenate(negs) == rec['negatives']).all(), 'negative ids differ'
    errs['loss'] = np.max(np.abs(np.concatenate(losses) - rec['losses']) / np.abs(rec['losses']))
    fr = {}
    for t in range(4):
        ref = rec['final_%d' % t]
        errs['final_%d' % t] = rel_inf(o.p[t].reshape(ref.shape), ref)
        fr['final_%d' % t] = frac_outside(o.p[t].reshape(ref.shape), ref)
    st = rng.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    po = bloom_oracle_for(case, rec, which='final')
    errs['predict_all'] = rel_inf(po.predict(3), rec['predict_user3_all'])
    errs['predict_pairs'] = rel_inf(po.predict(rec['predict_pairs_u'], rec['predict_pairs_i']), rec['predict_pairs'])
    return errs, fr


def replay_explicit_with_oracle(case, rec):
    """Replays a recorded ExplicitFactorizationModel run (factorization/explicit.py:173-284)."""
    loss = str(case['loss'])
    hp = oracle_hparams(case)
    mk = lambda: BilinearOracle(rec['init_0'], rec['init_1'], rec['init_2'], rec['init_3'],
                                opt=ORACLE_OPT[str(case['opt'])], **hp)
    o = mk()
    rng = Rng(state=('MT19937', rec['rng_key_before_fit'], int(rec['rng_pos_before_fit'])))
    N, B = int(case['N']), int(case['B'])
    users64, items64 = rec['users'].astype(np.int64), rec['items'].astype(np.int64)
    losses, errs = [], {}
    for e in range(int(case['n_iter'])):
        perm = rng.shuffle_perm(N)  # one permutation for the three arrays (torch_utils.py:35-52)
        su, si, sr = users64[perm], items64[perm], rec['ratings'][perm]
        assert (su == rec['shuffled_users'][e]).all() and (si == rec['shuffled_items'][e]).all()
        assert (sr == rec['shuffled_ratings'][e]).all()
        if e == 0:
            B0 = min(B, N)
            l0, dg = mk().explicit_step(su[:B0], si[:B0], sr[:B0], loss=loss, want_grads=True)
            errs['loss0'] = abs(l0 - rec['losses'][0]) / abs(rec['losses'][0])
            bscale = max(np.abs(rec['grad0_2']).max(), np.abs(rec['grad0_3']).max())
            for t in range(4):
                ref = rec['grad0_%d' % t]
                errs['grad0_%d' % t] = (rel_inf(dg[t].reshape(ref.shape), ref) if t < 2
                                        else np.abs(dg[t].reshape(ref.shape) - ref).max() / bscale)
        losses.append(o.explicit_train(su, si, sr, B, loss=loss))
    errs['loss'] = np.max(np.abs(np.concatenate(losses) - rec['losses']) / np.abs(rec['losses']))
    fr = {}
    for t in range(4):
        ref = rec['final_%d' % t]
        errs['final_%d' % t] = rel_inf(o.p[t].reshape(ref.shape), ref)
        fr['final_%d' % t] = frac_outside(o.p[t].reshape(ref.shape), ref)
    st = rng.get_state()
    assert (st[1] == rec['rng_key_after_fit']).all() and st[2] == int(rec['rng_pos_after_fit'])
    po = BilinearOracle(rec['final_0'], rec['final_1'], rec['final_2'], rec['final_3'])
    errs['predict_all'] = rel_inf(po.explicit_predict(3, None, loss=loss), rec['predict_all'])
    errs['predict_pairs'] = rel_inf(po.explicit_predict(rec['predict_users'], rec['predict_items'], loss=loss),
                                    rec['predict_pairs'])
    return errs, fr


def conditioned_trajectory_bounds(case, rec, rel_delta=1e-5):
    """How far a recorded multi-step run may legitimately be from an implementation whose SUMMED GRADIENTS agree with the
    reference's to `rel_delta` (north star: "fp32 loss/grad within 1e-5 rel"): the recorded run is replayed through the
    oracle one minibatch at a time and, per element, the first-order effect of a gradient perturbation of
    delta = rel_delta * ||g||inf (per embedding table; the two bias tables against their joint norm; touched rows only)
    on that step's update is accumulated over the steps:

        Adagrad (sparse or dense, +wd)    lr * delta / (sqrt(sum_pre + g^2) + eps)
        Adam / SparseAdam                 (lr / bc1) * delta / (sqrt(v_new / bc2) + eps)    (the moment carries the error on)

    The bound is only large where an accumulator is still ~0 and the summed gradient is itself ~0 (first steps: the
    update is lr * sign(g)), i.e. it singles out the ill-conditioned elements by gradient magnitude instead of allowing a
    blanket fraction of outliers.  Returns (oracle final tables, [bound per table]), tables shaped like the fixture's."""
    opt = ORACLE_OPT[case['opt']]
    hp = oracle_hparams(case)
    o = BilinearOracle(rec['init_0'], rec['init_1'], rec['init_2'], rec['init_3'], opt=opt, **hp)
    nn = int(case.get('n_neg', 5)) if case['loss'] == 'adaptive_hinge' else 1
    N, B = int(case['N']), int(case['B'])
    lr = float(hp.get('lr', 1e-2))
    wd = float(hp.get('weight_decay', 0.0))
    eps = 1e-10 if opt.startswith('adagrad') else 1e-8
    b1, b2 = 0.9, 0.999
    bounds = [np.zeros(o.p[t].shape, np.float64) for t in range(4)]
    step = 0
    for e in range(int(case['n_iter'])):
        su, si = rec['shuffled_users'][e].astype(np.int64), rec['shuffled_items'][e].astype(np.int64)
        for off in range(0, N, B):
            hi = min(off + B, N)
            neg = rec['negatives'][(e * N + off) * nn:(e * N + hi) * nn]
            pre_p = [x.copy() for x in o.p]
            pre_s1 = [x.copy() for x in o.s1]
            pre_s2 = [x.copy() for x in o.s2]
            _, g = o.step(su[off:hi], si[off:hi], neg, loss=case['loss'], n_neg=nn, want_grads=True)
            step += 1
            bscale = max(np.abs(g[2]).max(), np.abs(g[3]).max())
            for t in range(4):
                gt = g[t].astype(np.float64)
                scale = np.abs(gt).max() if t < 2 else bscale
                touched = (gt != 0).any(axis=1, keepdims=True) if gt.ndim == 2 else (gt != 0)
                delta = rel_delta * scale * touched
                geff = gt + wd * pre_p[t].astype(np.float64) if opt.endswith('dense') else gt
                if opt.startswith('adagrad'):
                    bounds[t] += lr * delta / (np.sqrt(pre_s1[t].astype(np.float64) + geff * geff) + eps)
                else:
                    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
                    v_new = b2 * pre_s2[t].astype(np.float64) + (1.0 - b2) * geff * geff
                    bounds[t] += (lr / bc1) * delta / (np.sqrt(v_new / bc2) + eps)
    finals = [o.p[t].reshape(rec['final_%d' % t].shape) for t in range(4)]
    return finals, [bounds[t].reshape(rec['final_%d' % t].shape) for t in range(4)]
