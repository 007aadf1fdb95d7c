#!/usr/bin/env python
"""Host-side timeline of ONE epoch of the serial fit() loop at C2 scale (what ImplicitFactorizationModel.fit does per epoch),
every step followed by a device synchronisation so that its GPU time is attributed to it.
usage: python scripts/trace_fit_serial.py [n_interactions]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
U, I = 10_000_000, 1_000_000
rs = np.random.RandomState(0)
inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=1 << 20, use_cuda=True, sparse=True,
                                   optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2), random_state=np.random.RandomState(1))
model.fit(inter)
torch.cuda.synchronize()
dev = model._net.tables()[0].device
eng, stream = host._engine_for(dev), host._stream_for(dev)
tables, binding = model._slk_tables(), model._bind()
rstate = np.random.RandomState(5)
out = {}


def timed(name, fn):
    t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[name] = {'host_ms': (t1 - t0) * 1e3, 'with_device_ms': (t2 - t0) * 1e3}
    return r


for rep in range(2):
    du0 = timed('ids_to_device users', lambda: host.ids_to_device(inter.user_ids, dev))
    di0 = timed('ids_to_device items', lambda: host.ids_to_device(inter.item_ids, dev))
    du, di = torch.empty_like(du0), torch.empty_like(di0)
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    mb = torch.empty((n + (1 << 20) - 1) >> 20, dtype=torch.float32, device=dev)
    timed('bilinear_reserve', lambda: eng.bilinear_reserve(tables, binding.as_struct(), n, 1 << 20, 'bpr', 1, stream=stream))
    timed('rng_set_state', lambda: eng.rng_set_state(rstate.get_state()))
    timed('shuffle_perm', lambda: eng.shuffle_perm(n, perm.data_ptr(), stream=stream))
    timed('gather users', lambda: eng.gather_rows_i64(du0.data_ptr(), perm.data_ptr(), n, 1, du.data_ptr(), stream=stream))
    timed('gather items', lambda: eng.gather_rows_i64(di0.data_ptr(), perm.data_ptr(), n, 1, di.data_ptr(), stream=stream))
    o = binding.as_struct()
    timed('bilinear_train', lambda: eng.bilinear_train(tables, o, du.data_ptr(), di.data_ptr(), n, 1 << 20, 'bpr', 1, mb.data_ptr(),
                                                       stream=stream))
    timed('rng_get_state', lambda: rstate.set_state(eng.rng_get_state()))
    timed('epoch loss read-back', lambda: float(mb.double().mean().item()))
print(json.dumps({'interactions': n, 'steps': out, 'sum_with_device_ms': sum(v['with_device_ms'] for v in out.values())}))
