"""Generate tests/golden/enc_*.npz by running the LIVE reference's ImplicitSequenceModel with the
LSTM / CNN / mixture-of-tastes representations (/root/reference, CPU; sequence/representations.py:147-596).

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_encoders.py

These encoders' bodies run on stock PyTorch in both implementations; what the fixtures pin is everything
around them: parameter creation order / initialisation under a seed, the RandomState consumption (shuffles,
negatives), the embedding front-end (lookup forward, gradient of every table row after the first minibatch,
incl. BloomEmbedding item layers and sparse=True layers), the per-minibatch losses and the trained parameters.
Recorded per case: every named parameter initially / its gradient at the first step / finally, per-epoch
shuffled sequences, per-minibatch negatives and losses, the RandomState after fit, two predict() calls."""
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
from spotlight.layers import BloomEmbedding  # noqa: E402
from spotlight.sequence.representations import CNNNet, LSTMNet, MixtureLSTMNet  # noqa: E402

from oracle.make_golden import optimizer_factory  # noqa: E402
from oracle.make_golden_seq import make_sequences  # noqa: E402



def build_representation(case):
    """None: let the model build it from the string (the reference's own construction path)."""
    kind, kw = case['rep'], dict(case.get('rep_kw', {}))
    if not kw and not case.get('bloom'):
        return kind
    if case.get('bloom'):
        kw['item_embedding_layer'] = BloomEmbedding(case['I'], case['D'], compression_ratio=case['ratio'],
                                                    num_hash_functions=int(case['bloom']), padding_idx=0)
    cls = {'lstm': LSTMNet, 'cnn': CNNNet, 'mixture': MixtureLSTMNet}[kind]
    return cls(case['I'], embedding_dim=case['D'], **kw)


def run_reference(case):
    rs = np.random.RandomState(case['data_seed'])
    seqs = make_sequences(rs, case['N'], case['L'], case['I'], case.get('pad_frac', 0.5))
    inter = SequenceInteractions(seqs, num_items=case['I'])
    model_rs = np.random.RandomState(case['seed'])
    # the constructor seeds torch from the RandomState; an explicitly built representation is created
    # AFTER it, as a user following the reference's examples would (examples/bloom_embeddings)
    model = ref_seq.ImplicitSequenceModel(
        loss=case['loss'], representation='pooling', embedding_dim=case['D'], n_iter=case['n_iter'],
        batch_size=case['B'], l2=case.get('l2', 0.0), learning_rate=case.get('lr', 1e-2),
        optimizer_func=optimizer_factory(case['opt']), sparse=case['opt'] == 'adagrad_sparse',
        random_state=model_rs, num_negative_samples=case.get('n_neg', 5))
    model._representation = build_representation(case)
    model._initialize(inter)
    params = dict(model._net.named_parameters())
    names = list(params)
    rec = {'names': np.array(names)}
    for t, nm in enumerate(names):
        rec['init_%d' % t] = params[nm].detach().numpy().copy()
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
    for t, nm in enumerate(names):
        rec['grad0_%d' % t] = first_grads[t]
        rec['final_%d' % t] = params[nm].detach().numpy().copy()
    rec['rng_key_after_fit'] = model_rs.get_state()[1].copy()
    rec['rng_pos_after_fit'] = np.int64(model_rs.get_state()[2])
    rec['predict_seq'] = seqs[1].copy()
    rec['predict_all'] = model.predict(seqs[1])
    pi = np.arange(1, min(case['I'], 12), dtype=np.int64).reshape(-1, 1)
    rec['predict_items'] = pi
    rec['predict_some'] = model.predict(seqs[2], pi)
    rec['predict_seq2'] = seqs[2].copy()
    for k, v in case.items():
        if k != 'rep_kw':
            rec['case_' + k] = np.array(v)
    for k, v in case.get('rep_kw', {}).items():
        rec['repkw_' + k] = np.array(v)
    return rec


def cases():
    base = dict(I=40, N=50, L=7, D=8, B=16, n_iter=2, seed=42, data_seed=7, l2=1e-6, lr=1e-2, n_neg=3)
    out = [
        dict(base, name='enc_lstm_bpr_adam_default', rep='lstm', loss='bpr', opt='adam_default'),
        dict(base, name='enc_lstm_hinge_adagrad_sparse', rep='lstm', loss='hinge', opt='adagrad_sparse', l2=0.0),
        dict(base, name='enc_lstm_adaptive_hinge_adagrad', rep='lstm', loss='adaptive_hinge', opt='adagrad', l2=0.0),
        dict(base, name='enc_cnn_pointwise_adam_default', rep='cnn', loss='pointwise', opt='adam_default'),
        dict(base, name='enc_cnn_deep_bpr_adagrad', rep='cnn', loss='bpr', opt='adagrad', l2=0.0,
             rep_kw=dict(kernel_width=(3, 2), dilation=(1, 2), num_layers=2, nonlinearity='relu',
                         residual_connections=True)),
        dict(base, name='enc_cnn_plain_hinge_adagrad', rep='cnn', loss='hinge', opt='adagrad', l2=0.0,
             rep_kw=dict(kernel_width=5, dilation=1, num_layers=1, nonlinearity='tanh', residual_connections=False)),
        dict(base, name='enc_mixture_bpr_adam_default', rep='mixture', loss='bpr', opt='adam_default'),
        dict(base, name='enc_mixture_adaptive_hinge_adagrad', rep='mixture', loss='adaptive_hinge', opt='adagrad',
             l2=0.0, rep_kw=dict(num_mixtures=3)),
        dict(base, name='enc_lstm_bloom_bpr_adagrad', rep='lstm', loss='bpr', opt='adagrad', l2=0.0, I=60, bloom=2,
             ratio=0.4),
        dict(base, name='enc_cnn_bloom_pointwise_adam_default', rep='cnn', loss='pointwise', opt='adam_default', I=60,
             bloom=3, ratio=0.5),
        dict(name='enc_lstm_d32_bpr_adam_default', rep='lstm', loss='bpr', opt='adam_default', I=300, N=120, L=20, D=32,
             B=32, n_iter=2, seed=1, data_seed=0, pad_frac=0.3, l2=0.0, lr=1e-2, n_neg=3),
    ]
    return out


def main():
    os.makedirs(golden_dir(), exist_ok=True)
    torch.set_num_threads(1)
    for case in cases():
        rec = run_reference(case)
        np.savez_compressed(os.path.join(golden_dir(), case['name'] + '.npz'), **rec)
        print('%-40s %d params, %d minibatches, loss %.4f -> %.4f' % (case['name'], len(rec['names']), len(rec['losses']),
                                                                    rec['losses'][0], rec['losses'][-1]))


if __name__ == '__main__':
    run_main(main)
