"""Generate tests/golden/explicit_*.npz by running the LIVE reference's ExplicitFactorizationModel
(/root/reference, CPU) and pin oracle/slk_oracle.c's restatement (slko_explicit_*) against it.

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_explicit.py

Recorded per case: initial parameters, per-epoch shuffled ids/ratings, per-minibatch losses, the
gradients of the first minibatch, final parameters / optimizer state / RandomState and predictions
(spotlight/factorization/explicit.py:173-284)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.reference_import import golden_dir, import_reference, run_main  # noqa: E402

import_reference()  # `spotlight` = the reference itself, never this repository's alias package onto the product

import spotlight.factorization.explicit as ref_exp  # noqa: E402
from spotlight.interactions import Interactions  # noqa: E402

from oracle.make_golden import optimizer_factory  # noqa: E402
from oracle.replay import replay_explicit_with_oracle  # noqa: E402

NAMES = ['user_embeddings.weight', 'item_embeddings.weight', 'user_biases.weight', 'item_biases.weight']


def make_data(case):
    rs = np.random.RandomState(case['data_seed'])
    users = rs.randint(0, case['U'], case['N']).astype(np.int32)
    items = rs.randint(0, case['I'], case['N']).astype(np.int32)
    if case['loss'] == 'logistic':
        ratings = rs.choice([-1.0, 1.0], case['N']).astype(np.float32)
    elif case['loss'] == 'poisson':
        ratings = rs.poisson(2.0, case['N']).astype(np.float32)
    else:
        ratings = rs.randint(1, 6, case['N']).astype(np.float32)
    return users, items, ratings


def run_reference(case):
    users, items, ratings = make_data(case)
    inter = Interactions(users, items, ratings=ratings, num_users=case['U'], num_items=case['I'])
    model_rs = np.random.RandomState(case['seed'])
    model = ref_exp.ExplicitFactorizationModel(
        loss=case['loss'], embedding_dim=case['D'], n_iter=case['n_iter'], batch_size=case['B'],
        l2=case.get('l2', 0.0), learning_rate=case.get('lr', 1e-2), optimizer_func=optimizer_factory(case['opt']),
        sparse=case['opt'] in ('adagrad_sparse', 'sparse_adam'), random_state=model_rs)
    model._initialize(inter)
    params = dict(model._net.named_parameters())
    rec = {'init_%d' % t: params[nm].detach().numpy().copy() for t, nm in enumerate(NAMES)}
    rec['rng_key_before_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_before_fit'] = np.int64(model_rs.get_state()[2])
    shuffled, losses, first_grads = [], [], []
    orig_shuffle = ref_exp.shuffle

    def rec_shuffle(*arrays, **kw):
        out = orig_shuffle(*arrays, **kw)
        shuffled.append([np.asarray(a).copy() for a in out])
        return out

    orig_loss = model._loss_func

    def rec_loss(*a, **kw):
        out = orig_loss(*a, **kw)
        losses.append(float(out.item()))
        return out

    orig_step = model._optimizer.step

    def rec_step(*a, **kw):
        if not first_grads:
            for nm in NAMES:
                g = params[nm].grad
                first_grads.append((g.to_dense() if g.is_sparse else g).detach().numpy().copy())
        return orig_step(*a, **kw)

    ref_exp.shuffle = rec_shuffle
    model._loss_func = rec_loss
    model._optimizer.step = rec_step
    try:
        model.fit(inter)
    finally:
        ref_exp.shuffle = orig_shuffle
    rec['users'], rec['items'], rec['ratings'] = users, items, ratings
    rec['shuffled_users'] = np.stack([s[0] for s in shuffled])
    rec['shuffled_items'] = np.stack([s[1] for s in shuffled])
    rec['shuffled_ratings'] = np.stack([s[2] for s in shuffled])
    rec['losses'] = np.array(losses, dtype=np.float32)
    st = model._optimizer.state
    for t, nm in enumerate(NAMES):
        rec['grad0_%d' % t] = first_grads[t]
        rec['final_%d' % t] = params[nm].detach().numpy().copy()
        s = st[params[nm]]
        if 'sum' in s:
            rec['state1_%d' % t] = s['sum'].detach().numpy().copy()
        else:
            rec['state1_%d' % t] = s['exp_avg'].detach().numpy().copy()
            rec['state2_%d' % t] = s['exp_avg_sq'].detach().numpy().copy()
    rec['rng_key_after_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_after_fit'] = np.int64(model_rs.get_state()[2])
    rec['predict_all'] = model.predict(3)
    pu, pi = np.arange(0, 12, dtype=np.int64) % case['U'], (np.arange(0, 12, dtype=np.int64) * 5 + 1) % case['I']
    rec['predict_users'], rec['predict_items'] = pu, pi
    rec['predict_pairs'] = model.predict(pu, pi)
    for k, v in case.items():
        rec['case_' + k] = np.array(v)
    return rec


def cases():
    out = []
    for loss in ('regression', 'poisson', 'logistic'):
        for opt in ('adam_default', 'adagrad', 'sparse_adam'):
            out.append(dict(name='explicit_%s_%s' % (loss, opt), loss=loss, opt=opt, U=30, I=40, N=400, D=8, B=64,
                            n_iter=2, seed=42, data_seed=7, l2=1e-6 if opt == 'adam_default' else 0.0, lr=1e-2))
    out.append(dict(name='explicit_d64_regression_adagrad', loss='regression', opt='adagrad', U=300, I=200, N=3000,
                    D=64, B=256, n_iter=2, seed=1, data_seed=0, frac_tol=0.15))
    return out


def main():
    os.makedirs(golden_dir(), exist_ok=True)
    torch.set_num_threads(1)
    for case in cases():
        rec = run_reference(case)
        errs, fr = replay_explicit_with_oracle(case, rec)
        step_keys = [k for k in errs if k.startswith('grad0') or k == 'loss0']
        m_step = max(errs[k] for k in step_keys)
        print('%-34s single-step err %.2e | trajectory err %.2e (%s) | frac outside %.3f'
              % (case['name'], m_step, max(errs.values()), max(errs, key=errs.get), max(fr.values())))
        assert m_step < 1e-5, errs
        assert errs['loss'] < 1e-3 and max(fr.values()) <= case.get('frac_tol', 0.05), (errs, fr)
        np.savez_compressed(os.path.join(golden_dir(), case['name'] + '.npz'), **rec)
    print('all explicit cases pinned')


if __name__ == '__main__':
    run_main(main)
