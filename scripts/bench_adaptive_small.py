#!/usr/bin/env python
"""adaptive hinge (num_negative_samples 5) at the reference's own operating points: engine time per minibatch at the C1 shape
and a mid-sized shape, minibatch 256 / 1024 / 4096, with the item side re-sorted per minibatch after the selection ('late')
or all 1+n occurrences sorted once per chunk.   usage: python scripts/bench_adaptive_small.py
`--routes`: instead, the persistent kernel (score phase + in-phase selection, csrc/slk_epoch.hip) against the per-minibatch
launches (chunk-sorted occurrences) at minibatch 256 / 1024 / 4096."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native  # noqa: E402

dev = torch.device('cuda', 0)
st = torch.cuda.current_stream(dev).cuda_stream
out = []
for shape, U, I, D in (('c1', 943, 1682, 32), ('mid', 1_000_000, 100_000, 64)):
    routes = '--routes' in sys.argv
    for B in ((256, 512, 1024, 2048) if routes else (256, 1024, 4096, 16384, 65536)):
        for late_min in ((-1, -2) if routes else (0, 1 << 40)):
            eng = _native.Engine(0)
            for kv in [x[len('--set='):] for x in sys.argv if x.startswith('--set=')]:  # e.g. --set=epoch_debug=8 (anatomy runs)
                eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
            if routes:
                eng.set_option('adaptive_late_min_batch', 1 << 40)
                eng.set_option('epoch_kernel', 1 if late_min == -1 else 0)
                eng.set_option('epoch_max_batch', 4096)
                eng.set_option('epoch_adaptive_max_batch', 4096)
            else:
                eng.set_option('adaptive_late_min_batch', late_min)
            gen = torch.Generator(device=dev)
            gen.manual_seed(3)
            tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen),
                      torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
                      torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
            s1 = [torch.zeros_like(t) for t in tables]
            tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D)
            op = _native.make_optim('adagrad', [t.data_ptr() for t in s1], None, lr=1e-2)
            K = max(16, min(400, (1 << 22) // B))
            n = 2 * K * B
            users = torch.randint(0, U, (n,), device=dev, generator=gen)
            items = torch.randint(0, I, (n,), device=dev, generator=gen)
            mb = torch.zeros(2 * K, device=dev)
            eng.rng_set_state(np.random.RandomState(1).get_state())
            run = lambda a: eng.bilinear_train(tb, op, users[a * B:].data_ptr(), items[a * B:].data_ptr(), K * B, B,
                                               'adaptive_hinge', 5, mb[a:].data_ptr(), stream=st)
            run(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out.append({'shape': shape, 'users': U, 'items': I, 'dim': D, 'batch': B, 'late_item_sort': late_min == 0,
                        'us_per_minibatch': dt / K * 1e6, 'interactions_per_s': K * B / dt})
            if routes:
                out[-1]['route'] = 'persistent kernel' if late_min == -1 else 'launches'
                del out[-1]['late_item_sort']
            eng.close()
            print(json.dumps(out[-1]), flush=True)
