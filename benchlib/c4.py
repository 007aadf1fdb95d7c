"""--workload c4: BASELINE.json configs[3], the PoolNet sequence step."""
import json
import time

import numpy as np
import torch

from benchlib.common import HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS
from spotlight_amd import _native


def bench_c4(args):
    """BASELINE.json configs[3]: ImplicitSequenceModel PoolNet, synthetic sequences len=200, 1M
    items, dim=64, bpr, Adagrad.  A step = one minibatch of `--batch` sequences; the unit is a
    (sequence, timestep) pair (32*D + 40 algorithmic bytes each).  Diagnostic
    workload, not the headline metric: prints its own JSON line."""
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    I, D, L, K, W = args.items, args.dim, args.seq_len, args.steps, args.warmup
    B = args.batch if args.batch != (1 << 20) else 4096
    eng = _native.Engine(0)
    eng.set_option('overlap_prep', 1)  # as fit() sets it on its ctx: the next chunk's negatives + sorts beside the passes
    for kv in args.set:
        name, value = kv.split('=')
        eng.set_option(name, int(value))
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    E = torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen)
    E[0] = 0
    bias = torch.zeros(I, device=dev)
    s1 = [torch.zeros_like(E), torch.zeros_like(bias)]
    tb = _native.make_seq_tables(E.data_ptr(), bias.data_ptr(), I, D)
    op = _native.make_optim('adagrad', [None, s1[0].data_ptr(), None, s1[1].data_ptr()], None, lr=1e-2)
    seqs = torch.randint(1, I, ((W + 2 * K) * B, L), device=dev, dtype=torch.int64, generator=gen)  # W warm-up + K timed + K profiled
    pad = float(getattr(args, 'pad_frac', 0.0) or 0.0)
    if pad > 0:
        # "+ a 20 %-left-padded variant" -- every sequence gets a left padding whose length is uniform in
        # [0, 2 * pad * L], so that `pad` of all positions are padding (id 0) and the sequences are ragged, as to_sequence's are
        npad = torch.randint(0, int(2 * pad * L) + 1, (seqs.shape[0],), device=dev, generator=gen)
        seqs[torch.arange(L, device=dev)[None, :] < npad[:, None]] = 0
    mb_loss = torch.zeros(W + 2 * K, device=dev)
    eng.rng_set_state(np.random.RandomState(5).get_state())
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(first, n_mb):
        eng.poolnet_train(tb, op, 0, seqs[first * B:].data_ptr(), n_mb * B, L, B, 'bpr', 1,
                          mb_loss[first:].data_ptr(), stream=stream)
    eng.poolnet_reserve(tb, op, K * B, L, B, 'bpr', 1, stream=stream)  # scratch of the timed call's shape
    run(0, W)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(W, K)  # timed region: no instrumentation inside
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # per-kernel durations: K more steps with hipEvents around every launch (the records cost ~10 us per launch and, around
    # the host-side parts of a chunk's preparation, also count the host's time)
    eng.profile_reset()
    eng.profile_enable(True)
    run(W + K, K)
    torch.cuda.synchronize(dev)
    eng.profile_enable(False)
    prof = eng.profile_read()
    ts = int((seqs[W * B:(W + K) * B] != 0).sum().item())  # the unit: non-padding (sequence, timestep) pairs = mask.sum() (sequence/implicit.py:250)
    alg = 32 * D + 40
    kern = {k: {'launches': prof[k][0], 'avg_ms': prof[k][1] / max(prof[k][0], 1)} for k in ('seq_pass', 'item_pass', 'epoch')
            if prof[k][0] or k != 'epoch'}  # 'epoch': minibatches of a few thousand timesteps run inside k_poolnet_epoch, one launch per chunk
    out = {'metric': 'training (sequence, timestep) pairs/sec, PoolNet BPR dim=%d' % D, 'value': ts / elapsed,
           'unit': 'timesteps/s', 'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3,
           'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'C4: PoolNet, %d sequences x len %d per minibatch, %d items, dim %d, bpr, '
                                  'adagrad, %s' % (B, L, I, D, 'no padding' if pad <= 0 else
                                                   '%g of the positions left padding (length uniform in [0, %d] per sequence); '
                                                   'unit = non-padding timesteps' % (pad, int(2 * pad * L)))},
           'roofline': {'bound': 'hbm', 'alg_bytes_per_timestep': alg, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'step_achieved': ts * alg / elapsed / 1e9, 'step_frac_of_peak': ts * alg / elapsed / 1e9 / HBM_PEAK_GBS,
                        'kernels': kern,
                        'other_ms_per_step': {k: prof[k][1] / K for k in ('sample', 'prep')}},
           'final_minibatch_loss': float(mb_loss[W + K - 1].item())}
    return out
