"""Times slk_sample_items alone (GPU): `count` draws in [0, num_items), device-synchronised wall clock around the call
(min of the repetitions); --out appends one JSON line per count.

    python scripts/bench_sampler.py [--out gpurun_out/x/bench_sampler.jsonl] [--set mt_long_min_blocks=...]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spotlight_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--out', default='')
ap.add_argument('--items', type=int, default=10 ** 6)
ap.add_argument('--counts', type=int, nargs='*', default=[1 << 20, 1 << 23, 1 << 25, 5 << 23])
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--set', action='append', default=[])
args = ap.parse_args()
eng = _native.Engine(0)
for kv in args.set:
    k, v = kv.split('=')
    eng.set_option(k, int(v))
eng.rng_set_state(np.random.RandomState(3).get_state())
stream = torch.cuda.current_stream().cuda_stream
for count in args.counts:
    out = torch.empty(count, dtype=torch.int64, device='cuda')
    ts = []
    for rep in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sample_items(args.items, count, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    rec = {'count': count, 'num_items': args.items, 'ms_min': min(ts) * 1e3, 'ms_all': [round(t * 1e3, 4) for t in ts],
           'G_draws_per_s': count / min(ts) / 1e9, 'options': args.set}
    print(json.dumps(rec), flush=True)
    if args.out:
        with open(args.out, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    del out
