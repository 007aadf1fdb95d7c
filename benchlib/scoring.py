"""--workload predict / eval: the far side of predict()."""
import json
import time

import numpy as np
import torch

from benchlib.common import HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS
from spotlight_amd import _native


def bench_scoring(args):
    """The far side of predict() at the C2 table sizes (BASELINE.md's predict roofline row).
    --workload predict: ImplicitFactorizationModel.predict(user) -- one user against every item; a step = one call;
    algorithmic bytes = items * (4 D + 4) read + items * 4 written.
    --workload eval: evaluation.mrr_score's device side -- `--batch` users (default 4096) with one held-out item each ranked
    against every item (slk_bilinear_rank: no score matrix); a step = one call; the unit is a (user, item) score; bound by
    the matrix cores (exact-fp32 MFMA: 2 D flop per score against the 157 TFLOP/s f32 MFMA peak).
    Diagnostic workloads, not the headline metric: each prints its own JSON line."""
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    U, I, D, K, W = args.users, args.items, args.dim, args.steps, args.warmup
    eng = _native.Engine(0)
    for kv in args.set:
        name, value = kv.split('=')
        eng.set_option(name, int(value))
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen), torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.empty(U, device=dev).normal_(0, 0.01, generator=gen), torch.empty(I, device=dev).normal_(0, 0.01, generator=gen)]
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.workload == 'predict':
        users = torch.randint(0, U, (W + K,), device=dev, dtype=torch.int64, generator=gen)
        out = torch.empty(I, device=dev)

        def step(k):
            eng.bilinear_predict(tb, users[k:].data_ptr(), 1, None, I, out.data_ptr(), stream)
        for k in range(W):
            step(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record()
        for k in range(W, W + K):
            step(k)
        ev1.record()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) / K
        alg = I * (4 * D + 4) + I * 4
        rec = {'metric': 'predict(user) calls/sec, all %d items, dim=%d' % (I, D), 'value': K / elapsed, 'unit': 'calls/s',
               'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'dtype': 'f32',
               'data': 'synthetic', 'vs_baseline': None,
               'config': {'workload': 'predict: one user against %d items, dim %d (C2 item table)' % (I, D)},
               'roofline': {'bound': 'hbm', 'kernel': 'k_score_rows<1>', 'alg_bytes_per_call': alg, 'device_ms_per_call': dev_ms,
                            'achieved': alg / dev_ms / 1e6, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg / dev_ms / 1e6 / HBM_PEAK_GBS,
                            'calls_per_s_at_peak': HBM_PEAK_GBS * 1e9 / alg, 'traffic': None}}
    else:
        R = args.batch if args.batch != (1 << 20) else 4096
        users = torch.randint(0, U, (R,), device=dev, dtype=torch.int64, generator=gen)
        row_group = torch.arange(R, device=dev, dtype=torch.int64)
        targets = torch.randint(0, I, (R,), device=dev, dtype=torch.int64, generator=gen)
        ranks = torch.empty(R, dtype=torch.float64, device=dev)

        def step():
            eng.bilinear_rank(tb, users.data_ptr(), R, row_group.data_ptr(), targets.data_ptr(), R, None, None, ranks.data_ptr(), stream)
        for _ in range(W):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(K):
            step()
        ev1.record()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) / K
        flop = 2.0 * D * R * I
        rec = {'metric': 'ranked (user, item) scores/sec, mrr_score device side, dim=%d' % D, 'value': R * I * K / elapsed,
               'unit': 'scores/s', 'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
               'dtype': 'f32', 'data': 'synthetic', 'vs_baseline': None,
               'config': {'workload': 'eval: %d users x 1 held-out item each ranked against %d items, dim %d (slk_bilinear_rank)' % (R, I, D)},
               'roofline': {'bound': 'mfma', 'kernel': 'k_score_gemm<2, COUNT>', 'flop_per_call': flop, 'device_ms_per_call': dev_ms,
                            'achieved': flop / dev_ms / 1e9, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': flop / dev_ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                            'item_table_bytes_streamed_per_call': ((R + 63) // 64) * I * (4 * D + 4), 'traffic': None},
               'mean_reciprocal_rank': float((1.0 / ranks).mean().item())}
    eng.close()
    return rec
