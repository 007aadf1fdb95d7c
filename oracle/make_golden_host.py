"""Generate tests/golden/host_api.npz from the LIVE reference (/root/reference): outputs of the
host-side callers around the training path for fixed seeds -- cross_validation splits, the synthetic
sequential generator, to_sequence, and evaluation metrics over a fixed score function.

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_host.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.reference_import import golden_dir, import_reference, run_main  # noqa: E402

import_reference()  # `spotlight` = the reference itself, never this repository's alias package onto the product

from spotlight.cross_validation import (random_train_test_split, shuffle_interactions,  # noqa: E402
                                        user_based_train_test_split)
from spotlight.datasets.synthetic import generate_sequential  # noqa: E402
from spotlight.evaluation import (mrr_score, precision_recall_score, rmse_score, sequence_mrr_score,  # noqa: E402
                                  sequence_precision_recall_score)


class FixedScores(object):
    """A 'model' whose predict() is a fixed function of the ids (so that the metrics are pinned
    independently of any training)."""

    def __init__(self, num_users, num_items, seed):
        rs = np.random.RandomState(seed)
        self.table = rs.normal(size=(num_users, num_items)).astype(np.float32)
        self.table[:, ::7] = 0.25  # ties
        self._num_items = num_items

    def predict(self, user_ids, item_ids=None):
        if item_ids is None:
            if np.ndim(user_ids) == 0:
                return self.table[int(user_ids)].copy()
            return self.table[int(np.asarray(user_ids).sum()) % len(self.table)].copy()  # a sequence
        return self.table[np.asarray(user_ids).reshape(-1), np.asarray(item_ids).reshape(-1)].copy()


def main():
    rec = {}
    data = generate_sequential(num_users=30, num_items=60, num_interactions=900, concentration_parameter=0.1, order=3,
                               random_state=np.random.RandomState(11))
    rec['gen_users'], rec['gen_items'], rec['gen_ts'], rec['gen_ratings'] = (data.user_ids, data.item_ids,
                                                                             data.timestamps, data.ratings)
    sh = shuffle_interactions(data, random_state=np.random.RandomState(12))
    rec['shuffle_users'], rec['shuffle_items'] = sh.user_ids, sh.item_ids
    tr, te = random_train_test_split(data, test_percentage=0.25, random_state=np.random.RandomState(13))
    rec['rs_train_users'], rec['rs_train_items'], rec['rs_test_users'], rec['rs_test_items'] = (
        tr.user_ids, tr.item_ids, te.user_ids, te.item_ids)
    utr, ute = user_based_train_test_split(data, test_percentage=0.3, random_state=np.random.RandomState(14))
    rec['us_train_users'], rec['us_test_users'], rec['us_test_items'] = utr.user_ids, ute.user_ids, ute.item_ids
    seq = te.to_sequence(max_sequence_length=6, min_sequence_length=2, step_size=2)
    rec['to_seq'] = seq.sequences
    model = FixedScores(30, 60, 15)
    rec['mrr'] = mrr_score(model, te)
    rec['mrr_train'] = mrr_score(model, te, train=tr)
    rec['seq_mrr'] = sequence_mrr_score(model, seq)
    rec['seq_mrr_excl'] = sequence_mrr_score(model, seq, exclude_preceding=True)
    p, r = precision_recall_score(model, te, train=tr, k=5)
    rec['prec5'], rec['rec5'] = p, r
    p, r = precision_recall_score(model, te, k=np.array([1, 3, 10]))
    rec['prec_multi'], rec['rec_multi'] = p, r
    p, r = sequence_precision_recall_score(model, seq, k=2, exclude_preceding=True)
    rec['seq_prec2'], rec['seq_rec2'] = p, r
    rec['rmse'] = np.float64(rmse_score(model, te))
    # end-to-end, the level the reference's own tests work at (tests/factorization/test_implicit.py,
    # tests/sequence/test_sequence_implicit.py: MRR floors): train the reference's models on the synthetic
    # data above and record the MRRs a drop-in run with the same seeds should reproduce
    import torch
    from spotlight.factorization.implicit import ImplicitFactorizationModel
    from spotlight.sequence.implicit import ImplicitSequenceModel
    torch.set_num_threads(1)
    big = generate_sequential(num_users=60, num_items=200, num_interactions=6000, concentration_parameter=0.01,
                              order=2, random_state=np.random.RandomState(21))
    btr, bte = random_train_test_split(big, test_percentage=0.2, random_state=np.random.RandomState(22))
    fm = ImplicitFactorizationModel(loss='bpr', embedding_dim=16, n_iter=4, batch_size=256, learning_rate=1e-2,
                                    l2=1e-6, random_state=np.random.RandomState(23))
    fm.fit(btr)
    rec['e2e_mrr_factorization'] = mrr_score(fm, bte, train=btr)
    str_, ste = user_based_train_test_split(big, test_percentage=0.3, random_state=np.random.RandomState(24))
    sq_tr = str_.to_sequence(max_sequence_length=10, min_sequence_length=3, step_size=1)
    sq_te = ste.to_sequence(max_sequence_length=10, min_sequence_length=3, step_size=1)
    sm = ImplicitSequenceModel(loss='bpr', representation='pooling', embedding_dim=16, n_iter=4, batch_size=64,
                               learning_rate=1e-2, l2=1e-6, random_state=np.random.RandomState(25))
    sm.fit(sq_tr)
    rec['e2e_mrr_sequence'] = sequence_mrr_score(sm, sq_te)
    np.savez_compressed(os.path.join(golden_dir(), 'host_api.npz'), **rec)
    print('host_api.npz written:', {k: np.asarray(v).shape for k, v in rec.items()})


if __name__ == '__main__':
    run_main(main)
