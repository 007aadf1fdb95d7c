#!/usr/bin/env python
"""End-to-end epoch time of ImplicitSequenceModel with the torch-side encoders (LSTM / CNN), embedding lookups
through this package's front-end (csrc/slk_embed.hip) vs the same model with the lookups on torch's own
embedding ops (what the reference does on PyTorch-ROCm).  The encoder body (MIOpen) is identical in both.
usage: bench_encoders.py [items] [dim] [batch] [seq_len] [sequences]   prints one JSON line (GRAFT_OUT: file)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import embedding, layers  # noqa: E402
from spotlight_amd.interactions import SequenceInteractions  # noqa: E402
from spotlight_amd.sequence.implicit import ImplicitSequenceModel  # noqa: E402

I = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
L = int(sys.argv[4]) if len(sys.argv) > 4 else 50
N = int(sys.argv[5]) if len(sys.argv) > 5 else 32768
rs = np.random.RandomState(0)
seqs = np.minimum(rs.zipf(1.3, (N, L)), I - 1).astype(np.int32)
data = SequenceInteractions(seqs, num_items=I)


def torch_lookup(weight, ids, bloom=None, padding_idx=None, sparse=False):
    assert bloom is None
    return torch.nn.functional.embedding(ids, weight, padding_idx=padding_idx, sparse=sparse)


def epoch_seconds(rep, front_end):
    layers.lookup = embedding.lookup if front_end else torch_lookup
    model = ImplicitSequenceModel(loss='bpr', representation=rep, embedding_dim=D, batch_size=B, n_iter=1,
                                  learning_rate=1e-3, use_cuda=True, random_state=np.random.RandomState(1))
    model.fit(data)  # warm-up epoch (MIOpen find, allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(data)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


out = {'items': I, 'dim': D, 'batch': B, 'seq_len': L, 'sequences': N, 'optimizer': 'dense Adam (reference default)'}
for rep in ('lstm', 'cnn'):
    ours, stock = epoch_seconds(rep, True), epoch_seconds(rep, False)
    out[rep] = {'epoch_s_front_end': ours, 'epoch_s_torch_embedding': stock,
                'timesteps_per_s_front_end': N * L / ours, 'timesteps_per_s_torch_embedding': N * L / stock}
layers.lookup = embedding.lookup
line = json.dumps(out)
print(line)
if os.environ.get('GRAFT_OUT'):
    os.makedirs(os.path.dirname(os.path.abspath(os.environ['GRAFT_OUT'])), exist_ok=True)
    with open(os.environ['GRAFT_OUT'], 'w') as f:
        f.write(line + '\n')
