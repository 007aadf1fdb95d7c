"""Generate tests/golden/seq_*.npz by running the LIVE reference's ImplicitSequenceModel with
PoolNet (/root/reference, CPU) and pin oracle/slk_oracle.c's PoolNet restatement against it.

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_seq.py

Recorded per case: initial parameters, per-epoch shuffled sequences, per-minibatch negatives and
losses, the gradients of the first minibatch, final parameters / optimizer state / RandomState
and a few predictions (spotlight/sequence/implicit.py:193-340)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.reference_import import golden_dir, import_reference, run_main  # noqa: E402

import_reference()  # `spotlight` = the reference itself, never this repository's alias package onto the product

import spotlight.sequence.implicit as ref_seq  # noqa: E402
from spotlight.interactions import SequenceInteractions  # noqa: E402

from oracle.make_golden import optimizer_factory  # noqa: E402
from oracle.replay import replay_seq_with_oracle  # noqa: E402

NAMES = ['item_embeddings.weight', 'item_biases.weight']


def make_sequences(rs, n_seq, L, num_items, pad_frac):
    seqs = rs.randint(1, num_items, (n_seq, L)).astype(np.int32)
    for b in range(n_seq):
        if rs.rand() < pad_frac:
            k = rs.randint(1, L)  # left padding, at least one real item
            seqs[b, :k] = 0
    return seqs


def run_reference(case):
    rs = np.random.RandomState(case['data_seed'])
    seqs = make_sequences(rs, case['N'], case['L'], case['I'], case.get('pad_frac', 0.5))
    inter = SequenceInteractions(seqs, num_items=case['I'])
    model_rs = np.random.RandomState(case['seed'])
    names = list(NAMES)
    representation = 'pooling'
    if case.get('bloom'):
        # PoolNet over a BloomEmbedding item layer (sequence/representations.py:62-68, layers.py:74-244)
        from spotlight.layers import BloomEmbedding
        from spotlight.sequence.representations import PoolNet
        torch.manual_seed(int(case['seed']))
        representation = PoolNet(case['I'], case['D'], item_embedding_layer=BloomEmbedding(
            case['I'], case['D'], compression_ratio=case['ratio'], num_hash_functions=int(case['bloom']),
            padding_idx=0))
        names[0] = 'item_embeddings.embeddings.weight'
    model = ref_seq.ImplicitSequenceModel(
        loss=case['loss'], representation=representation, embedding_dim=case['D'], n_iter=case['n_iter'],
        batch_size=case['B'], l2=case.get('l2', 0.0), learning_rate=case.get('lr', 1e-2),
        optimizer_func=optimizer_factory(case['opt']),
        sparse=case['opt'] in ('adagrad_sparse', 'sparse_adam'),
        random_state=model_rs, num_negative_samples=case.get('n_neg', 5))
    model._initialize(inter)
    params = dict(model._net.named_parameters())
    rec = {'init_%d' % t: params[nm].detach().numpy().copy() for t, nm in enumerate(names)}
    rec['rng_key_before_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_before_fit'] = np.int64(model_rs.get_state()[2])
    shuffled, negatives, losses, first_grads = [], [], [], []
    orig_shuffle, orig_sample = ref_seq.shuffle, ref_seq.sample_items

    def rec_shuffle(*arrays, **kw):
        out = orig_shuffle(*arrays, **kw)
        shuffled.append(np.asarray(out).copy())
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
            for nm in names:
                g = params[nm].grad
                first_grads.append((g.to_dense() if g.is_sparse else g).detach().numpy().copy())
        return orig_step(*a, **kw)

    ref_seq.shuffle, ref_seq.sample_items = rec_shuffle, rec_sample
    model._loss_func = rec_loss
    model._optimizer.step = rec_step
    try:
        model.fit(inter)
    finally:
        ref_seq.shuffle, ref_seq.sample_items = orig_shuffle, orig_sample
    rec['sequences'] = seqs
    rec['shuffled'] = np.stack(shuffled)
    rec['negatives'] = np.concatenate(negatives)
    rec['losses'] = np.array(losses, dtype=np.float32)
    st = model._optimizer.state
    for t, nm in enumerate(names):
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
    rec['predict_seq'] = seqs[1].copy()
    rec['predict_all'] = model.predict(seqs[1])
    pi = np.arange(1, min(case['I'], 12), dtype=np.int64).reshape(-1, 1)
    rec['predict_items'] = pi
    rec['predict_some'] = model.predict(seqs[2], pi)
    rec['predict_seq2'] = seqs[2].copy()
    for k, v in case.items():
        rec['case_' + k] = np.array(v)
    return rec


def cases():
    out = []
    for loss in ('pointwise', 'bpr', 'hinge', 'adaptive_hinge'):
        for opt in ('adam_default', 'adagrad', 'adagrad_sparse', 'sparse_adam'):
            out.append(dict(name='seq_%s_%s' % (loss, opt), loss=loss, opt=opt, I=40, N=50, L=7, D=8, B=16,
                            n_iter=2, seed=42, data_seed=7, l2=1e-6, lr=1e-2, n_neg=3))
    out.append(dict(name='seq_d64_bpr_adagrad', loss='bpr', opt='adagrad', I=300, N=120, L=20, D=64, B=32,
                    n_iter=2, seed=1, data_seed=0, pad_frac=0.3, frac_tol=0.15))
    out.append(dict(name='seq_d32_adaptive_adam', loss='adaptive_hinge', opt='adam_default', I=200, N=100, L=10,
                    D=32, B=256, n_iter=2, seed=2, data_seed=3, l2=1e-6, n_neg=5, pad_frac=0.0, frac_tol=0.15))
    return out


def bloom_cases():
    """PoolNet over a BloomEmbedding item layer (dense gradients only: the inner table is not sparse)."""
    out = []
    for loss, opt in (('bpr', 'adagrad'), ('pointwise', 'adam_default'), ('hinge', 'adagrad'),
                      ('adaptive_hinge', 'adagrad')):
        out.append(dict(name='seq_bloom_%s_%s' % (loss, opt), loss=loss, opt=opt, I=60, N=50, L=7, D=8, B=16,
                        n_iter=2, seed=42, data_seed=7, l2=1e-6 if opt == 'adam_default' else 0.0, lr=1e-2,
                        n_neg=3, bloom=2, ratio=0.4))
    out.append(dict(name='seq_bloom_d64_bpr_adagrad', loss='bpr', opt='adagrad', I=400, N=120, L=20, D=64, B=32,
                    n_iter=2, seed=1, data_seed=0, pad_frac=0.3, frac_tol=0.15, bloom=4, ratio=0.2))
    return out


def main():
    os.makedirs(golden_dir(), exist_ok=True)
    torch.set_num_threads(1)
    which = bloom_cases() if 'bloom' in sys.argv[1:] else cases()
    for case in which:
        rec = run_reference(case)
        errs, fr = replay_seq_with_oracle(case, rec)
        step_keys = [k for k in errs if k.startswith('grad0') or k == 'loss0']
        m_step = max(errs[k] for k in step_keys)
        print('%-34s single-step err %.2e | trajectory err %.2e (%s) | frac outside %.3f'
              % (case['name'], m_step, max(errs.values()), max(errs, key=errs.get), max(fr.values())))
        assert m_step < 1e-5, errs
        # trajectories are only conditionally stable (see oracle/make_golden.py): Adagrad's first
        # step is lr*sign(g), and the bias gradient of an item that is a positive target in one
        # timestep and a negative in another is cancellation noise at initialisation
        assert errs['loss'] < 1e-3 and max(fr.values()) <= case.get('frac_tol', 0.05), (errs, fr)
        np.savez_compressed(os.path.join(golden_dir(), case['name'] + '.npz'), **rec)
    print('all sequence cases pinned')


if __name__ == '__main__':
    run_main(main)
