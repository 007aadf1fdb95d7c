"""Record the LIVE reference's mean held-out MRR for every case of tests/reference_floors.py (the reference's own
tests/sequence/test_sequence_implicit.py configurations) -> tests/golden/reference_floors.json.

TEST INFRASTRUCTURE.  Run in the build container only:   python oracle/make_golden_floors.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle.reference_import import import_reference  # noqa: E402

import_reference()  # `spotlight` = the reference itself, never this repository's alias package onto the product

import reference_floors as rf  # noqa: E402
from spotlight.cross_validation import user_based_train_test_split  # noqa: E402
from spotlight.datasets import synthetic  # noqa: E402
from spotlight.evaluation import sequence_mrr_score  # noqa: E402
from spotlight.layers import BloomEmbedding  # noqa: E402
from spotlight.sequence.implicit import ImplicitSequenceModel  # noqa: E402
from spotlight.sequence.representations import CNNNet, LSTMNet, MixtureLSTMNet, PoolNet  # noqa: E402


def reference_mrr(case):
    name, rep, rep_kw, bloom, model_kw, interactions, concentration, floor = case
    rs = np.random.RandomState(rf.SEED)
    data = synthetic.generate_sequential(num_users=100, num_items=100, num_interactions=interactions,
                                         concentration_parameter=concentration, order=2, random_state=rs)
    train, test = user_based_train_test_split(data, random_state=rs)
    train = train.to_sequence(max_sequence_length=10, step_size=None)
    test = test.to_sequence(max_sequence_length=10, step_size=None)
    representation = rep
    if rep_kw or bloom:
        kw = dict(rep_kw)
        if bloom:
            kw['item_embedding_layer'] = BloomEmbedding(train.num_items, rf.DIM, compression_ratio=bloom[0],
                                                        num_hash_functions=bloom[1])
        cls = {'pooling': PoolNet, 'lstm': LSTMNet, 'cnn': CNNNet, 'mixture': MixtureLSTMNet}[rep]
        representation = cls(train.num_items, embedding_dim=rf.DIM, **kw)
    kw = dict(loss='bpr', batch_size=rf.BATCH, embedding_dim=rf.DIM)
    kw.update(model_kw)
    model = ImplicitSequenceModel(representation=representation, random_state=rs, **kw)
    model.fit(train)
    return float(sequence_mrr_score(model, test).mean())


if __name__ == '__main__':
    torch.set_num_threads(1)
    out = {}
    for case in rf.CASES:
        out[case[0]] = reference_mrr(case)
        assert out[case[0]] > case[-1], case[0]
        print('%-30s floor %.2f reference %.4f' % (case[0], case[-1], out[case[0]]))
    with open(os.path.join(ROOT, 'tests', 'golden', 'reference_floors.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
