#!/usr/bin/env python
"""Interactions.to_sequence: device route (slk_seqprep.hip) at n interactions vs the host double loop
(interactions.py:170-266) on a bounded sample.  usage: bench_to_sequence.py [n] [max_len] [step]
Prints one JSON line; with GRAFT_OUT set also writes it there."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
step = int(sys.argv[3]) if len(sys.argv) > 3 else 200
U, I = max(n // 400, 10), 1_000_000
rs = np.random.RandomState(0)
users = rs.randint(0, U, n).astype(np.int32)
items = rs.randint(1, I, n).astype(np.int32)
ts = rs.randint(0, 1 << 30, n).astype(np.int32)

dev = torch.device('cuda', 0)
eng = _native.Engine(0)
stream = torch.cuda.current_stream(dev).cuda_stream
d_u = torch.from_numpy(users).to(dev).to(torch.int64)
d_i = torch.from_numpy(items).to(dev).to(torch.int64)
d_t = torch.from_numpy(ts).to(dev).to(torch.int64)
best = None
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = eng.to_sequence_plan(d_u.data_ptr(), d_i.data_ptr(), d_t.data_ptr(), 0, n, U, L, step, 1, stream)
    seq = torch.empty((rows, L), dtype=torch.int32, device=dev)
    su = torch.empty((rows,), dtype=torch.int32, device=dev)
    eng.to_sequence_fill(seq.data_ptr(), su.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)

# whole drop-in call (upload + kernels + download of the int32 matrix)
inter = Interactions(users, items, timestamps=ts, num_users=U, num_items=I)
t0 = time.perf_counter()
s_dev = inter.to_sequence(max_sequence_length=L, step_size=step, device='cuda')
t_call = time.perf_counter() - t0

# host route on a sample of whole users (same density), extrapolated
m = min(n, 2_000_000)
keep = users < max(int(U * (m / n)), 1)
sample = Interactions(users[keep], items[keep], timestamps=ts[keep], num_users=U, num_items=I)
t0 = time.perf_counter()
s_host = sample.to_sequence(max_sequence_length=L, step_size=step)
t_host = time.perf_counter() - t0
k = s_host.sequences.shape[0]
assert np.array_equal(s_dev.sequences[:k], s_host.sequences) and np.array_equal(s_dev.user_ids[:k], s_host.user_ids)

out = {'n': n, 'max_sequence_length': L, 'step_size': step, 'num_users': U, 'sequences': int(rows),
       'device_kernels_ms': best * 1e3, 'device_call_ms_with_pcie': t_call * 1e3,
       'host_sample_interactions': int(keep.sum()), 'host_sample_ms': t_host * 1e3,
       'host_extrapolated_ms': t_host * 1e3 * n / max(int(keep.sum()), 1),
       'bytes_out': int(rows) * L * 4, 'sample_rows_checked_equal': int(k)}
line = json.dumps(out)
print(line)
if os.environ.get('GRAFT_OUT'):
    os.makedirs(os.path.dirname(os.path.abspath(os.environ['GRAFT_OUT'])), exist_ok=True)
    with open(os.environ['GRAFT_OUT'], 'w') as f:
        f.write(line + '\n')
