#!/usr/bin/env python
"""Where does the first overlapped training call of a process lose ~2 ms?  Times three K-step calls after different warm-ups."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native

mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
eng = _native.Engine(0)
U, I, D, B, K, W = 10_000_000, 1_000_000, 64, 1 << 20, 20, 5
gen = torch.Generator(device=dev); gen.manual_seed(1)
tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen), torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
          torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
s1 = [torch.zeros_like(t) for t in tables]
tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D)
op = _native.make_optim('adagrad', [t.data_ptr() for t in s1], None, lr=1e-2)
n_total = (W + K) * B
users = torch.randint(0, U, (n_total,), device=dev, dtype=torch.int64, generator=gen)
items = torch.randint(0, I, (n_total,), device=dev, dtype=torch.int64, generator=gen)
mb = torch.zeros(W + K, device=dev)
side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev)); torch.cuda.set_stream(side)
stream = torch.cuda.current_stream(dev).cuda_stream if mode != 'nullstream' else 0
def run(first, n_mb):
    eng.bilinear_train(tb, op, users[first * B:].data_ptr(), items[first * B:].data_ptr(), n_mb * B, B, 'bpr', 1, mb[first:].data_ptr(), stream=stream)
if mode in ('inline', 'busy', 'busy1s', 'w5x2', 'legsfirst', 'idle'):
    eng.set_option('overlap_prep', 0)
eng.rng_set_state(np.random.RandomState(1).get_state())
eng.bilinear_reserve(tb, op, K * B, B, 'bpr', 1, stream=stream)
run(0, W)
if mode == 'tiny':      # a tiny overlapped call first: 2 chunks of one minibatch
    eng.set_option('chunk_interactions', 1 << 20)
    run(0, 2)
    eng.set_option('chunk_interactions', 1 << 23)
elif mode == 'w9':      # warm-up long enough to be overlapped itself
    run(0, 9)
elif mode == 'w5x2':    # the same 5-minibatch warm-up twice
    run(0, W)
elif mode == 'busy1s':  # ~1 s of copy kernels
    n = 1 << 28
    bufs = [torch.empty(n, device=dev) for _ in range(3)]
    for _ in range(30):
        eng.probe_stream(0, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), n, iters=10, stream=stream)
elif mode == 'legsfirst':  # bench.py's secondary legs (two overlapped K-step calls) BEFORE the timed in-line call
    eng.set_option('overlap_prep', 1)
    run(W, K); run(W, K)
    torch.cuda.synchronize(dev)
    eng.set_option('overlap_prep', 0)
    run(0, W)
elif mode == 'busy':    # ~100 ms of copy kernels right before the timed calls (clock / power state, not the ids)
    n = 1 << 28
    bufs = [torch.empty(n, device=dev) for _ in range(3)]
    for _ in range(3):
        eng.probe_stream(0, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), n, iters=10, stream=stream)
torch.cuda.synchronize(dev)
out = []
for j in range(4 if mode == 'idle' else 3):
    if mode == 'idle' and j == 3:
        time.sleep(0.5)  # an idle GPU before the fourth call: does the first-call figure come back?
    t0 = time.perf_counter(); run(W, K); torch.cuda.synchronize(dev); out.append(round((time.perf_counter() - t0) / K * 1e3, 4))
print(mode, out)
