#!/usr/bin/env python
"""Same-process option sweep of the fused step (C2 shapes by default): the tables are built once, every configuration
(a set of slk_ctx_set_option values) runs W warm-up + K timed + K instrumented minibatches from the same ids.

    python scripts/sweep_engine.py --out gpurun_out/sweep.jsonl --configs 'overlap_prep=1,user_grid_mult=6' 'overlap_prep=2' ...

One JSON line per configuration: ms per step (wall clock around the timed call, stream synchronised), per-class kernel
times of the instrumented pass.  The first configuration is always the defaults, and it is repeated at the end (drift of
the box during the sweep shows up as the difference between the two)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spotlight_amd import _native  # noqa: E402

DEFAULTS = {'chunk_interactions': 1 << 23, 'overlap_prep': 0, 'overlap_min_batch': 1 << 16, 'item_grid_mult': 128,
            'user_grid_mult': 8, 'nt': 3, 'user_bias_zero_hint': 1, 'record_nt_min_bytes': 192 << 20, 'user_lat_max_batch': 1 << 17, 'item_long_gate': 1, 'item_lat_max_tiles': 2048}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=10_000_000)
    ap.add_argument('--items', type=int, default=1_000_000)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--batch', type=int, default=1 << 20)
    ap.add_argument('--steps', type=int, default=32)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--opt', default='adagrad')
    ap.add_argument('--loss', default='bpr')
    ap.add_argument('--user-zipf', type=float, default=0.0)
    ap.add_argument('--item-zipf', type=float, default=0.0)
    ap.add_argument('--repeat', type=int, default=2, help='timed calls per configuration (the minimum is reported too)')
    ap.add_argument('--out', default='')
    ap.add_argument('--configs', nargs='*', default=[])
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    eng = _native.Engine(0)
    U, I, D, B, K, W = args.users, args.items, args.dim, args.batch, args.steps, args.warmup
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
    s1 = [torch.zeros_like(t) for t in tables]
    s2 = [torch.zeros_like(t) for t in tables] if args.opt != 'adagrad' else None
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D, user_bias_zero=not bool(tables[2].any()))  # (as bench.py and fit() do)
    n_total = (W + K) * B
    users = torch.randint(0, U, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    items = torch.randint(0, I, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    def zipf_ids(n_ids, s_exp):
        w = 1.0 / torch.arange(1, n_ids + 1, device=dev, dtype=torch.float64) ** s_exp
        cdf = torch.cumsum(w / w.sum(), 0)
        ranks = torch.searchsorted(cdf, torch.rand(n_total, device=dev, dtype=torch.float64, generator=gen)).clamp_(max=n_ids - 1)
        return torch.randperm(n_ids, device=dev, generator=gen)[ranks]
    if args.item_zipf > 0:
        items = zipf_ids(I, args.item_zipf)
    if args.user_zipf > 0:
        users = zipf_ids(U, args.user_zipf)
    mb_loss = torch.zeros(W + K, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out_f = open(args.out, 'a') if args.out else None

    def run_config(label, opts):
        cfg = dict(DEFAULTS)
        cfg.update(opts)
        for k, v in cfg.items():
            eng.set_option(k, int(v))
        op = _native.make_optim(args.opt, [t.data_ptr() for t in s1], [t.data_ptr() for t in s2] if s2 else None, lr=1e-2)

        def run(first, n_mb):
            off = first * B
            eng.bilinear_train(tb, op, users[off:].data_ptr(), items[off:].data_ptr(), n_mb * B, B, args.loss, 1,
                               mb_loss[first:].data_ptr(), stream=stream)
        eng.rng_set_state(np.random.RandomState(1).get_state())
        eng.bilinear_reserve(tb, op, K * B, B, args.loss, 1, stream=stream)
        run(0, W)
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            run(W, K)
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / K * 1e3)
        eng.profile_reset()
        eng.profile_enable(True)
        run(W, K)
        torch.cuda.synchronize(dev)
        eng.profile_enable(False)
        prof = eng.profile_read()
        rec = {'label': label, 'opts': opts, 'ms_per_step': min(times), 'ms_per_step_all': times,
               'G_interactions_per_s': B / min(times) / 1e6,
               'class_ms_per_step': {k: prof[k][1] / K for k in ('sample', 'prep', 'user_pass', 'item_pass')},
               'loss_last': float(mb_loss[W + K - 1].item())}
        line = json.dumps(rec)
        print(line, flush=True)
        if out_f:
            out_f.write(line + '\n')
            out_f.flush()

    configs = [('defaults', {})]
    for c in args.configs:
        opts = {}
        for kv in c.split(','):
            if kv:
                k, v = kv.split('=')
                opts[k] = int(v)
        configs.append((c, opts))
    configs.append(('defaults (again)', {}))
    for label, opts in configs:
        try:
            run_config(label, opts)
        except Exception as e:  # one refused configuration must not lose the rest of the sweep
            line = json.dumps({'label': label, 'error': repr(e)[:300]})
            print(line, flush=True)
            if out_f:
                out_f.write(line + '\n')
                out_f.flush()
    eng.close()


if __name__ == '__main__':
    main()
