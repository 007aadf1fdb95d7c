#!/usr/bin/env python
"""Throughput of the explicit-feedback step (slk_bilinear_train_explicit) on C2-shaped tables: 10M users x 1M
items, dim 64, regression loss, Adagrad, batch 2^20, ids and ratings resident in HBM.  Diagnostic."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native  # noqa: E402

loss = sys.argv[1] if len(sys.argv) > 1 else 'regression'
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # 0: staged route (score pass + loss kernel before the user pass)
U, I, D, B, K, W = 10_000_000, 1_000_000, 64, 1 << 20, 16, 4
dev = torch.device('cuda', 0)
eng = _native.Engine(0)
eng.set_option('explicit_fused', fused)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen),
          torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
          torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
s1 = [torch.zeros_like(t) for t in tables]
tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D)
op = _native.make_optim('adagrad', [t.data_ptr() for t in s1], None, lr=1e-2)
n = (W + K) * B
users = torch.randint(0, U, (n,), device=dev, generator=gen)
items = torch.randint(0, I, (n,), device=dev, generator=gen)
ratings = (torch.randint(0, 2, (n,), device=dev, generator=gen).float() * 2 - 1) if loss == 'logistic' else \
    torch.randint(1, 6, (n,), device=dev, generator=gen).float()
mb = torch.zeros(W + K, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
run = lambda a, k: eng.bilinear_train_explicit(tb, op, users[a * B:].data_ptr(), items[a * B:].data_ptr(),
                                               ratings[a * B:].data_ptr(), k * B, B, loss, mb[a:].data_ptr(), stream=st)
eng.bilinear_reserve(tb, op, K * B, B, loss, 0, stream=st)
run(0, W)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(W, K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
eng.profile_reset()
eng.profile_enable(True)
run(W, K)
torch.cuda.synchronize()
eng.profile_enable(False)
prof = eng.profile_read()
print(json.dumps({'workload': 'explicit %s, %d users x %d items, dim %d, adagrad, batch %d' % (loss, U, I, D, B), 'fused': fused,
                  'interactions_per_s': K * B / dt, 'ms_per_step': dt / K * 1e3,
                  'ms_per_step_by_class': {k: prof[k][1] / K for k in ('prep', 'score', 'user_pass', 'item_pass')},
                  'final_loss': float(mb[-1].item())}))
