"""Generate tests/golden/bloom_*.npz: the LIVE reference's ImplicitFactorizationModel with a
BilinearNet whose embedding layers are BloomEmbeddings (spotlight/layers.py:74-244), and pin
oracle/slk_oracle.c's bloom restatement against it.

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_bloom.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.reference_import import golden_dir, import_reference, run_main  # noqa: E402

import_reference()  # `spotlight` = the reference itself, never this repository's alias package onto the product

import spotlight.factorization.implicit as ref_implicit  # noqa: E402
from spotlight.factorization.representations import BilinearNet  # noqa: E402
from spotlight.interactions import Interactions  # noqa: E402
from spotlight.layers import BloomEmbedding  # noqa: E402

from oracle.make_golden import optimizer_factory  # noqa: E402
from oracle.replay import replay_bloom_with_oracle  # noqa: E402

NAMES = ['user_embeddings', 'item_embeddings', 'user_biases', 'item_biases']


def weight_of(layer):
    return layer.embeddings.weight if isinstance(layer, BloomEmbedding) else layer.weight


def run_reference(case):
    rs = np.random.RandomState(case['data_seed'])
    users = rs.randint(0, case['U'], case['N']).astype(np.int32)
    items = rs.randint(0, case['I'], case['N']).astype(np.int32)
    inter = Interactions(users, items, num_users=case['U'], num_items=case['I'])
    model_rs = np.random.RandomState(case['seed'])
    model = ref_implicit.ImplicitFactorizationModel(
        loss=case['loss'], embedding_dim=case['D'], n_iter=case['n_iter'], batch_size=case['B'],
        l2=case.get('l2', 0.0), learning_rate=case.get('lr', 1e-2), optimizer_func=optimizer_factory(case['opt']),
        random_state=model_rs, num_negative_samples=case.get('n_neg', 5))
    # the layers are created AFTER the model's set_seed (as a user script would), in the order
    # user layer, item layer, then BilinearNet's own biases
    mk = lambda n: BloomEmbedding(n, case['D'], compression_ratio=case['ratio'],
                                  num_hash_functions=case['H'], bag=bool(case.get('bag', False)))
    ul = mk(case['U']) if case['user_bloom'] else None
    il = mk(case['I']) if case['item_bloom'] else None
    net = BilinearNet(case['U'], case['I'], case['D'], user_embedding_layer=ul, item_embedding_layer=il)
    model._representation = net
    model._initialize(inter)
    tabs = [weight_of(getattr(net, nm)) for nm in NAMES]
    rec = {'init_%d' % t: w.detach().numpy().copy() for t, w in enumerate(tabs)}
    rec['rng_key_before_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_before_fit'] = np.int64(model_rs.get_state()[2])
    shuffled, negatives, losses, first_grads = [], [], [], []
    orig_shuffle, orig_sample = ref_implicit.shuffle, ref_implicit.sample_items

    def rec_shuffle(*arrays, **kw):
        out = orig_shuffle(*arrays, **kw)
        shuffled.append([np.asarray(o).copy() for o in out])
        return out

    def rec_sample(*a, **kw):
        out = orig_sample(*a, **kw)
        negatives.append(np.asarray(out).copy().ravel())
        return out

    orig_loss = model._loss_func

    def rec_loss(*a, **kw):
        out = orig_loss(*a, **kw)
        losses.append(float(out.item()))
        return out

    orig_step = model._optimizer.step

    def rec_step(*a, **kw):
        if not first_grads:
            for w in tabs:
                g = w.grad
                first_grads.append((g.to_dense() if g.is_sparse else g).detach().numpy().copy())
        return orig_step(*a, **kw)

    ref_implicit.shuffle, ref_implicit.sample_items = rec_shuffle, rec_sample
    model._loss_func = rec_loss
    model._optimizer.step = rec_step
    try:
        model.fit(inter)
    finally:
        ref_implicit.shuffle, ref_implicit.sample_items = orig_shuffle, orig_sample
    rec['users'], rec['items'] = users, items
    rec['shuffled_users'] = np.stack([s[0] for s in shuffled])
    rec['shuffled_items'] = np.stack([s[1] for s in shuffled])
    rec['negatives'] = np.concatenate(negatives)
    rec['losses'] = np.array(losses, dtype=np.float32)
    st = model._optimizer.state
    for t, w in enumerate(tabs):
        rec['grad0_%d' % t] = first_grads[t]
        rec['final_%d' % t] = w.detach().numpy().copy()
        s = st[w]
        rec['state1_%d' % t] = (s['sum'] if 'sum' in s else s['exp_avg']).detach().numpy().copy()
    rec['rng_key_after_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_after_fit'] = np.int64(model_rs.get_state()[2])
    rec['predict_user3_all'] = model.predict(3)
    pu = np.arange(0, min(case['U'], 20), dtype=np.int64)
    pi = (pu * 7 + 1) % case['I']
    rec['predict_pairs_u'], rec['predict_pairs_i'] = pu, pi
    rec['predict_pairs'] = model.predict(pu, pi)
    for k, v in case.items():
        rec['case_' + k] = np.array(v)
    return rec


def cases():
    out = []
    base = dict(U=60, I=80, N=300, D=8, B=64, n_iter=2, seed=42, data_seed=7, l2=1e-6, lr=1e-2, n_neg=3,
                ratio=0.4, H=4)
    for loss, opt in (('bpr', 'adagrad'), ('adaptive_hinge', 'adam_default'), ('pointwise', 'adagrad'),
                      ('hinge', 'adam_default')):
        out.append(dict(base, name='bloom_item_%s_%s' % (loss, opt), loss=loss, opt=opt, user_bloom=0, item_bloom=1))
    out.append(dict(base, name='bloom_both_bpr_adagrad', loss='bpr', opt='adagrad', user_bloom=1, item_bloom=1))
    out.append(dict(base, name='bloom_both_adaptive_adam', loss='adaptive_hinge', opt='adam_default', user_bloom=1,
                    item_bloom=1, H=2))
    out.append(dict(base, name='bloom_user_pointwise_adagrad', loss='pointwise', opt='adagrad', user_bloom=1,
                    item_bloom=0))
    # bag=True is not recorded: the reference builds its EmbeddingBag offsets with stride 1
    # instead of num_hash_functions (layers.py:219-222), so a bag 'sum' is a single hashed row (and the
    # last bag swallows the tail) -- a quirk this package does not reproduce (BloomEmbedding(bag=True)
    # raises NotImplementedError in spotlight_amd).
    # C3-shaped: adaptive hinge n=5, bloom item table (compression 0.2, 4 hashes), D=128
    out.append(dict(name='bloom_c3_adaptive_adagrad', loss='adaptive_hinge', opt='adagrad', U=300, I=500, N=1200,
                    D=128, B=256, n_iter=2, seed=1, data_seed=0, n_neg=5, ratio=0.2, H=4, user_bloom=0,
                    item_bloom=1, frac_tol=0.15))
    return out


def main():
    os.makedirs(golden_dir(), exist_ok=True)
    torch.set_num_threads(1)
    for case in cases():
        rec = run_reference(case)
        errs, fr = replay_bloom_with_oracle(case, rec)
        step_keys = [k for k in errs if k.startswith('grad0') or k == 'loss0']
        m_step = max(errs[k] for k in step_keys)
        print('%-34s single-step err %.2e | trajectory err %.2e (%s) | frac outside %.3f'
              % (case['name'], m_step, max(errs.values()), max(errs, key=errs.get), max(fr.values())))
        assert m_step < 1e-5, errs
        assert errs['loss'] < 1e-3 and max(fr.values()) <= case.get('frac_tol', 0.05), (errs, fr)
        np.savez_compressed(os.path.join(golden_dir(), case['name'] + '.npz'), **rec)
    print('all bloom cases pinned')


if __name__ == '__main__':
    run_main(main)
