#!/usr/bin/env python
"""Host-side timeline of the pipelined epoch loop at C2 scale: how long the training call takes to ENQUEUE, how long the
next epoch's preparation takes beside it, and how long the epoch's kernels still run after that.
usage: python scripts/trace_fit_large.py [n_interactions]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization import implicit as host  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    U, I = 10_000_000, 1_000_000
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
    opt = lambda p: torch.optim.Adagrad(p, lr=1e-2)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=1 << 20, optimizer_func=opt,
                                       use_cuda=True, sparse=True, random_state=np.random.RandomState(1))
    model.fit(inter)
    torch.cuda.synchronize()
    dev = model._net.tables()[0].device
    eng, stream = host._engine_for(dev), host._stream_for(dev)
    prep, prep_stream = host._prep_lane_for(dev)
    tables, binding = model._slk_tables(), model._bind()
    du0, di0 = host.ids_to_device(inter.user_ids, dev), host.ids_to_device(inter.item_ids, dev)
    bufs = [(torch.empty_like(du0), torch.empty_like(di0), torch.empty(n, dtype=torch.int64, device=dev)) for _ in range(2)]
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    mb = torch.empty((n + (1 << 20) - 1) >> 20, dtype=torch.float32, device=dev)
    rstate = np.random.RandomState(5)

    def prepare(slot):
        prep.rng_set_state(rstate.get_state())
        u, i, g = bufs[slot]
        host.device_epoch_shuffle(prep, rstate, n, perm, [(du0, u, 1), (di0, i, 1)], prep_stream)
        prep.sample_items(I, n, g.data_ptr(), stream=prep_stream)
        rstate.set_state(prep.rng_get_state())

    out = []
    for mode in ('train only', 'prepare only', 'both'):
        prepare(0)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            t1 = t2 = t0
            if mode != 'prepare only':
                u, i, g = bufs[0]
                o = binding.as_struct()
                eng.bilinear_train(tables, o, u.data_ptr(), i.data_ptr(), n, 1 << 20, 'bpr', 1, mb.data_ptr(),
                                   d_neg_in=g.data_ptr(), stream=stream)
                t1 = time.perf_counter()
            if mode != 'train only':
                prepare(1)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
        out.append({'mode': mode, 'enqueue_train_ms': (t1 - t0) * 1e3, 'prepare_ms': (t2 - t1) * 1e3,
                    'drain_ms': (t3 - t2) * 1e3, 'total_ms': (t3 - t0) * 1e3})
    print(json.dumps({'interactions': n, 'timeline': out}))


if __name__ == '__main__':
    main()
