#!/usr/bin/env python
"""Embedding front-end (csrc/slk_embed.hip) vs torch's own embedding forward + backward on the device, at the
shapes the sequence encoders use: [batch, seq_len] lookups into an [items, dim] table; and a BloomEmbedding
lookup vs the reference's route (cached hash table -> index_select -> embedding -> sum, layers.py:177-242).
usage: bench_embedding.py [items] [dim] [batch] [seq_len]      prints one JSON line (GRAFT_OUT: also a file)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.embedding import lookup  # noqa: E402
from spotlight_amd.layers import BloomEmbedding  # noqa: E402

I = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
L = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device('cuda', 0)
rs = np.random.RandomState(0)


def timed(fn, reps=10):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {'items': I, 'dim': D, 'batch': B, 'seq_len': L, 'lookups': B * L}
w = torch.randn(I, D, device=dev).requires_grad_(True)
up = torch.randn(B, L, D, device=dev)
for dist in ('uniform', 'zipf'):
    ids_np = rs.randint(1, I, (B, L)) if dist == 'uniform' else np.minimum(rs.zipf(1.2, (B, L)), I - 1)
    ids = torch.from_numpy(ids_np).to(dev)

    def ours(sparse=False):
        w.grad = None
        (lookup(w, ids, padding_idx=0, sparse=sparse) * up).sum().backward()

    def theirs(sparse=False):
        w.grad = None
        (torch.nn.functional.embedding(ids, w, padding_idx=0, sparse=sparse) * up).sum().backward()

    def body_only():  # the multiply + sum + its backward that both variants share
        x = up.clone().requires_grad_(True)
        (x * up).sum().backward()

    out[dist] = {'front_end_dense_ms': timed(ours), 'torch_dense_ms': timed(theirs),
                 'front_end_sparse_ms': timed(lambda: ours(True)), 'torch_sparse_ms': timed(lambda: theirs(True)),
                 'shared_elementwise_ms': timed(body_only)}

# bloom layer: compression 0.2, 4 hashes
layer = BloomEmbedding(I, D, compression_ratio=0.2, num_hash_functions=4).to(dev)
ids = torch.from_numpy(rs.randint(1, I, (B, L))).to(dev)
from sklearn.utils import murmurhash3_32  # noqa: E402
t0 = time.perf_counter()
ar = np.arange(I, dtype=np.int32)
table = np.stack([murmurhash3_32(ar, seed=s) % layer.compressed_num_embeddings for s in layer._masks], 1).astype(np.int64)
table[0] = 0
hash_table = torch.from_numpy(table).to(dev)
out['bloom_reference_hash_table_build_ms'] = (time.perf_counter() - t0) * 1e3


def bloom_ours():
    layer.weight.grad = None
    (layer(ids) * up).sum().backward()


def bloom_reference_route():
    layer.weight.grad = None
    hashed = torch.index_select(hash_table, 0, ids.reshape(-1))
    emb = torch.nn.functional.embedding(hashed, layer.weight, padding_idx=0).sum(1).view(B, L, D)
    (emb * up).sum().backward()


out['bloom'] = {'front_end_ms': timed(bloom_ours), 'reference_route_ms': timed(bloom_reference_route)}
line = json.dumps(out)
print(line)
if os.environ.get('GRAFT_OUT'):
    os.makedirs(os.path.dirname(os.path.abspath(os.environ['GRAFT_OUT'])), exist_ok=True)
    with open(os.environ['GRAFT_OUT'], 'w') as f:
        f.write(line + '\n')
