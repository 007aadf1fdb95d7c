#!/usr/bin/env python
"""bench.py -- training interactions/sec of the fused BPR embedding step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one minibatch of the hot path (SURVEY.md 8(d), config C2 of BASELINE.json):
synthetic uniform ids over 10M users x 1M items, dim 64, bpr loss, Adagrad(lr=1e-2), batch
1,048,576 -- on-GPU numpy-exact negative draw, radix-sort grouping, user pass, item pass
(gather + dot + loss + backward + optimizer), all inside the timed region, ids resident in
HBM beforehand.  value = interactions processed by all ranks / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline     dominant kernel: algorithmic bytes per launch / mean launch
               duration from hipEvents recorded on the launch stream (slk_profile_*)
  cpu_baseline Spotlight's own CPU PyTorch path (oracle/_ref, staged by oracle/make_ref.sh) timed on the
               host cores on a bounded sample of the same workload (rank 0, N=1 only; kind "reference");
               oracle/slk_oracle.c (scalar C port, 1 thread) as the secondary field `port`
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from spotlight_amd import _native  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable


def algorithmic_bytes(dim, opt_state_words):
    """per interaction, fp32, int64 ids, no credit for cache hits/duplicates.
    user pass: user row param R+W + state R+W, two item rows read, user/pos/neg ids, user bias
    R+W(+state), two item-bias reads; item pass: two item rows written + state R+W, two item
    biases written + state R+W."""
    s = opt_state_words
    user_pass = (8 * dim + 8 * dim * s) + 2 * 4 * dim + 16 + (8 + 8 * s) + 8
    item_pass = 2 * (4 * dim + 8 * dim * s) + 2 * (4 + 8 * s)
    return user_pass, item_pass


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes over this very
    workload (profiles/pmc_traffic.json, produced by scripts/pmc_run.sh + scripts/summarize_pmc.py:
    2 x FETCH_SIZE + WRITE_SIZE, separate passes; MI355X_MICROARCH.md "HBM").  PMC counters cannot
    be collected from inside the benchmark process, so this is null unless the committed
    measurement matches the workload being run."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    cfg = rec.get('config', {})
    same = all(cfg.get(k) == getattr(args, k) for k in ('users', 'items', 'dim', 'batch', 'loss', 'opt'))
    return rec.get('kernels', {}).get(kernel, {}).get('hbm_bytes_per_launch') if same else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--users', type=int, default=None, help='user rows per GPU (default: the workload\'s)')
    ap.add_argument('--items', type=int, default=None, help='item rows per GPU (default: the workload\'s)')
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--batch', type=int, default=1 << 20)
    ap.add_argument('--loss', default='bpr')
    ap.add_argument('--opt', default='adagrad', choices=['adagrad', 'sparse_adam', 'adam_dense'],
                    help='adam_dense = the reference default: Adam(lr=1e-2, weight_decay=1e-6) over every row every step')
    ap.add_argument('--workload', default=None, choices=['c2', 'c3', 'c4', 'c5', 'predict', 'eval'],
                    help='c2: BilinearNet BPR step (the headline metric; the default at --gpus 1); c3: adaptive hinge n=5 over a '
                         'BloomEmbedding item table, dim 128; c4: PoolNet sequence step; c5: the per-GPU shard of the 1B-item x '
                         '100M-user table (12.5M users x 125M items per GPU; at --gpus 8 the full C5; the default at --gpus > 1: '
                         'the row-sharded configuration BASELINE.json names).  Explicit --users / --items override the shape.')
    ap.add_argument('--no-denominators', action='store_true',
                    help='N > 1: skip rank 0\'s world-1 runs of the same per-GPU shape (fused path, row-sharded path) that give '
                         'the scaling factor its stated denominators')
    ap.add_argument('--seq-len', type=int, default=200)
    ap.add_argument('--sharded', action='store_true',
                    help='run the row-sharded exchange path even at N=1 (diagnostic; default at N>1)')
    ap.add_argument('--slices', type=int, default=0, help='row-sharded path: user-slices per minibatch (0: default)')
    ap.add_argument('--item-zipf', type=float, default=0.0,
                    help='positive item ids drawn with probability ~ 1 / rank^s (s = this value; 0 = uniform, the default and the '
                         'metric\'s distribution) through a random rank -> id map: a stress run for duplicate handling')
    ap.add_argument('--user-zipf', type=float, default=0.0,
                    help='user ids drawn with probability ~ 1 / rank^s through a random rank -> id map (0 = uniform, the default and '
                         'the metric\'s distribution): power users, a stress run for the user pass\'s long-run form')
    ap.add_argument('--no-fit', action='store_true',
                    help='N=1, C2: skip the end-to-end ImplicitFactorizationModel.fit() measurement (the drop-in API around the engine)')
    ap.add_argument('--fit-interactions', type=int, default=1 << 25)
    ap.add_argument('--shard-chunk', type=int, default=8,
                    help='row-sharded path: minibatches per chunk (one count exchange + host synchronisation per chunk)')
    ap.add_argument('--side-stream', type=int, default=1, help='1: run the engine on a dedicated HIP stream')
    ap.add_argument('--set', action='append', default=[], metavar='NAME=VALUE',
                    help='engine tuning option (slk_ctx_set_option), e.g. item_grid_mult=28')
    ap.add_argument('--backend', default='hip', choices=['hip', 'emu'],
                    help="hip: the product (libspotlight_hip.so on MI355X, RCCL).  emu: TEST HARNESS ONLY -- the same host code over "
                         "tests/emu's CPU build of the kernels and gloo, so that the N > 1 launch path can be exercised on a box "
                         "without GPUs (tests/test_bench_cli.py); never a measurement")
    ap.add_argument('--no-probes', action='store_true', help='skip the copy / triad / step-ceiling bandwidth probes')
    ap.add_argument('--no-sharded-check', action='store_true',
                    help='N=1: skip the consistency run of the row-sharded path (world 1) against the fused path')
    ap.add_argument('--no-overlapped', action='store_true',
                    help='skip the secondary roofline.overlapped measurement (three more K-step calls with option overlap_prep = 1): what '
                         'the rocprofv3 / PMC commands use, so that their per-kernel means are those of the timed configuration only')
    ap.add_argument('--no-loss-check', action='store_true', help='measurement of debug modes whose results are meaningless')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=30.0)
    args = ap.parse_args()
    custom = args.users is not None or args.items is not None
    if args.workload is None:
        args.workload = 'c5' if (args.gpus > 1 and not custom) else 'c2'
    shape = {'c5': (12_500_000, 125_000_000)}.get(args.workload, (10_000_000, 1_000_000))
    args.users = args.users if args.users is not None else shape[0]
    args.items = args.items if args.items is not None else shape[1]
    return args


def reference_cpu_baseline(args, seconds):
    """Spotlight's own CPU PyTorch path (the copy staged by oracle/make_ref.sh under oracle/_ref/) timed
    on this machine's host cores by oracle/ref_cpu_baseline.py, in its own process: same table shapes, loss
    and minibatch as the GPU workload, protocol of the reference's examples/bloom_embeddings/performance.py:24-38.
    Returns None when the copy is not staged."""
    import subprocess
    script = os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py')
    if not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'spotlight')):
        return None
    cmd = [sys.executable, script, '--users', str(args.users), '--items', str(args.items), '--dim', str(args.dim),
           '--batch', str(args.batch), '--loss', args.loss, '--seconds', str(seconds)]
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60 + 12 * seconds)
        rec = json.loads(res.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
        return {'error': repr(e)[:300]}
    if 'sparse_adagrad' not in rec:
        return {'error': str(rec)[:300]}
    sa, da = rec['sparse_adagrad'], rec.get('default_dense_adam')
    out = {'value': sa['interactions_per_s'], 'unit': 'interactions/s', 'cores': sa['threads'], 'kind': 'reference',
           'cpu_model': rec['cpu_model'], 'host_cores': rec['host_cores'],
           'interactions_per_s_by_threads': sa['interactions_per_s_by_threads'],
           'sample': 'spotlight ImplicitFactorizationModel.fit() on CPU PyTorch %s, sparse=True + Adagrad(lr=1e-2), %s loss, '
                     '%d users x %d items, dim %d, minibatch %d (bounded sample; the GPU workload uses %d): warm-up fit + min '
                     'of 2 timed fits of %d minibatch(es) (%.1f s each); torch.set_num_threads: every host core (%d, on an eighth of a '
                     'minibatch) and 16 were probed, the faster (%d) was timed%s'
                     % (rec['torch'], rec['loss'], rec['users'], rec['items'], rec['dim'], rec['batch'], rec['gpu_workload_batch'],
                        sa['minibatches_per_fit'], sa['seconds'], rec['host_cores'], sa['threads'],
                        '; ' + rec['note'] if rec['note'] else '')}
    if da:
        out['reference_default_dense_adam'] = {'value': da['interactions_per_s'], 'unit': 'interactions/s', 'cores': da['threads'],
                                               'sample': '%d minibatch(es) per fit, %.1f s' % (da['minibatches_per_fit'], da['seconds'])}
    return out


def cpu_baseline(args, seconds):
    """The oracle (CPU port of spotlight/factorization/implicit.py:223-243 with a row-sparse
    Adagrad, i.e. the reference's sparse=True + Adagrad path) on the same table shapes."""
    from oracle.oracle import BilinearOracle, Rng
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 16 << 30
    U, I, D = args.users, args.items, args.dim
    need = (U + I) * D * 4 * 5
    note = ''
    while need > 0.6 * avail and U > 100_000:
        U //= 2
        need = (U + I) * D * 4 * 5
        note = ' (user table scaled to %d rows to fit host RAM)' % U
    rs = np.random.default_rng(0)
    block = (rs.standard_normal(1 << 20, dtype=np.float32) / D)
    p = [np.resize(block, (U, D)), np.resize(block, (I, D)), np.zeros(U, np.float32), np.zeros(I, np.float32)]
    ora = BilinearOracle(*p, opt=args.opt, lr=1e-2, sparse_grads=True)
    del p
    rng = Rng(seed=1)
    B = min(args.batch, 1 << 18)
    done, t_total = 0, 0.0
    # warm the page tables of the gradient buffers with one untimed minibatch
    users, items = rs.integers(0, U, B), rs.integers(0, I, B)
    ora.train(rng, users, items, B, loss=args.loss)
    while t_total < seconds and done < (1 << 24):
        users, items = rs.integers(0, U, B), rs.integers(0, I, B)
        t0 = time.perf_counter()
        ora.train(rng, users, items, B, loss=args.loss)
        t_total += time.perf_counter() - t0
        done += B
    return {'value': done / t_total, 'unit': 'interactions/s', 'cores': 1, 'kind': 'port',
            'sample': '%d minibatches of %d interactions, same table shapes%s, oracle/slk_oracle.c '
                      'single thread (%d host cores present)' % (done // B, B, note, os.cpu_count())}


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
        MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
        rec = {'metric': 'ranked (user, item) scores/sec, mrr_score device side, dim=%d' % D, 'value': R * I * K / elapsed,
               'unit': 'scores/s', 'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
               'dtype': 'f32', 'data': 'synthetic', 'vs_baseline': None,
               'config': {'workload': 'eval: %d users x 1 held-out item each ranked against %d items, dim %d (slk_bilinear_rank)' % (R, I, D)},
               'roofline': {'bound': 'mfma', 'kernel': 'k_score_gemm<2, COUNT>', 'flop_per_call': flop, 'device_ms_per_call': dev_ms,
                            'achieved': flop / dev_ms / 1e9, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': flop / dev_ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                            'item_table_bytes_streamed_per_call': ((R + 63) // 64) * I * (4 * D + 4), 'traffic': None},
               'mean_reciprocal_rank': float((1.0 / ranks).mean().item())}
    print(json.dumps(rec), flush=True)
    eng.close()


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
    ts = K * B * L
    alg = 32 * D + 40
    kern = {k: {'launches': prof[k][0], 'avg_ms': prof[k][1] / max(prof[k][0], 1)} for k in ('seq_pass', 'item_pass', 'epoch')
            if prof[k][0] or k != 'epoch'}  # 'epoch': minibatches of a few thousand timesteps run inside k_poolnet_epoch, one launch per chunk
    out = {'metric': 'training (sequence, timestep) pairs/sec, PoolNet BPR dim=%d' % D, 'value': ts / elapsed,
           'unit': 'timesteps/s', 'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3,
           'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'C4: PoolNet, %d sequences x len %d per minibatch, %d items, dim %d, bpr, '
                                  'adagrad, no padding' % (B, L, I, D)},
           'roofline': {'bound': 'hbm', 'alg_bytes_per_timestep': alg, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'step_achieved': ts * alg / elapsed / 1e9, 'step_frac_of_peak': ts * alg / elapsed / 1e9 / HBM_PEAK_GBS,
                        'kernels': kern,
                        'other_ms_per_step': {k: prof[k][1] / K for k in ('sample', 'prep')}},
           'final_minibatch_loss': float(mb_loss[W + K - 1].item())}
    print(json.dumps(out), flush=True)


def bench_c3(args):
    """BASELINE.json configs[2]: 10M users x 1M items, adaptive_hinge_loss n_neg=5, BloomEmbedding item
    table (compression 0.2 -> 200k rows, 4 hash functions), dim 128, Adagrad.  A step = one minibatch.
    Algorithmic bytes per interaction: 208*D + 90 (hashes computed in-kernel).
    Diagnostic workload, not the headline metric: prints its own JSON line."""
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    U, I, K, W = args.users, args.items, args.steps, args.warmup
    D = args.dim if args.dim != 64 else 128
    B = args.batch if args.batch != (1 << 20) else (1 << 18)
    NN, H = 5, 4
    rows = int(0.2 * I)
    eng = _native.Engine(0)
    eng.set_option('overlap_prep', 1)  # as fit() sets it on its ctx: the next chunk's negatives + sorts beside the passes
    for kv in args.set:
        name, value = kv.split('=')
        eng.set_option(name, int(value))
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.empty(rows, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
    tables[1][0] = 0  # padding row of the compressed table (layers.py:152-154)
    s1 = [torch.zeros_like(t) for t in tables]
    ib = _native.make_bloom(rows, H)
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D, item_bloom=ib)
    op = _native.make_optim('adagrad', [t.data_ptr() for t in s1], None, lr=1e-2)
    n_total = (W + K) * B
    users = torch.randint(0, U, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    items = torch.randint(0, I, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    mb_loss = torch.zeros(W + K, device=dev)
    eng.rng_set_state(np.random.RandomState(3).get_state())
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(first, n_mb):
        eng.bilinear_train(tb, op, users[first * B:].data_ptr(), items[first * B:].data_ptr(), n_mb * B, B,
                           'adaptive_hinge', NN, mb_loss[first:].data_ptr(), stream=stream)
    eng.bilinear_reserve(tb, op, K * B, B, 'adaptive_hinge', NN, stream=stream)
    run(0, W)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(W, K)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # kernel classes: a second, instrumented pass over the same minibatches' worth of work
    eng.profile_reset()
    eng.profile_enable(True)
    run(W, K)
    torch.cuda.synchronize(dev)
    eng.profile_enable(False)
    prof = eng.profile_read()
    alg = 208 * D + 90
    out = {'metric': 'training interactions/sec, adaptive hinge n=5, bloom item table, dim=%d' % D,
           'value': K * B / elapsed, 'unit': 'interactions/s', 'n_gpus': 1, 'steps': K, 'warmup': W,
           'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'C3: %d users x %d items, BloomEmbedding item table %d rows x %d hashes, dim %d, '
                                  'adaptive_hinge n_neg=%d, adagrad, minibatch %d; prep overlapped as fit() runs it' % (U, I, rows, H, D, NN, B)},
           'roofline': {'bound': 'hbm (92%% of the algorithmic bytes target a %d MB table + state that fit the '
                                 'Infinity Cache)' % (rows * D * 4 >> 20),
                        'alg_bytes_per_interaction': alg, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'step_achieved': K * B * alg / elapsed / 1e9,
                        'step_frac_of_peak': K * B * alg / elapsed / 1e9 / HBM_PEAK_GBS,
                        'ms_per_step_by_class': {k: prof[k][1] / K for k in
                                                 ('sample', 'prep', 'score', 'user_pass', 'item_pass')}},
           'final_minibatch_loss': float(mb_loss[-1].item())}
    print(json.dumps(out), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: launch the N ranks ourselves (one process
    per GPU, torch.distributed.run on 127.0.0.1) and pass rank 0's JSON line through.  Fails loudly when the
    machine does not have N GPUs -- it never degrades to fewer ranks."""
    import socket
    import subprocess
    if args.backend == 'hip':
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d requested but %d HIP device(s) visible; refusing to run fewer ranks\n'
                             % (args.gpus, have))
            return 3
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SLK_BENCH_SPAWNED='1')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


class Backend(object):
    """Device plumbing of the benchmark: 'hip' = torch-ROCm tensors + RCCL + libspotlight_hip.so (the product);
    'emu' = CPU tensors + gloo + tests/emu's build of the same kernels (test harness for the launch logic)."""

    def __init__(self, kind, local_rank):
        self.kind = kind
        if kind == 'hip':
            torch.cuda.set_device(local_rank)
            self.dev = torch.device('cuda', local_rank)
            self.engine = _native.Engine(local_rank)
            self.dist_backend = 'nccl'
            self.name = torch.cuda.get_device_name(local_rank)
        else:
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            from emu_backend import emu_lib
            self.dev = torch.device('cpu')
            self.engine = _native.Engine(0, lib=emu_lib())
            self.dist_backend = 'gloo'
            self.name = 'cpu emulator (test harness)'
        self.side = None

    def init_dist(self, rank, world, local_rank):
        """Process group + a self-check of everything the first multi-GPU run could trip over, BEFORE any table is allocated:
        every failure names the rank, the device and the variable to look at, and exits non-zero within the time-out instead of
        hanging (the driver's 8-GPU run is the first hardware run of this path: it must not be lost to a launcher problem)."""
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29400')
        who = 'bench.py rank %d/%d (local rank %d)' % (rank, world, local_rank)

        def die(code, msg):
            sys.stderr.write('%s: %s\n' % (who, msg))
            sys.stderr.flush()
            os._exit(code)
        timeout = datetime.timedelta(seconds=int(os.environ.get('SLK_BENCH_DIST_TIMEOUT', '180')))
        try:
            if self.kind == 'hip':
                if os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0') != '0':
                    die(4, 'HSA_ENABLE_IPC_MODE_LEGACY=%s: this host driver only supports dmabuf IPC; RCCL needs it unset or 0'
                        % os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])
                if torch.cuda.device_count() <= local_rank:
                    die(4, 'LOCAL_RANK %d but only %d HIP device(s) visible (HIP_VISIBLE_DEVICES=%s)'
                        % (local_rank, torch.cuda.device_count(), os.environ.get('HIP_VISIBLE_DEVICES')))
                os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')  # a failed collective raises instead of hanging
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank),
                                        timeout=timeout)
            else:
                dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timeout)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 -- rendezvous / RCCL initialisation
            die(4, 'init_process_group failed: %r (MASTER_ADDR=%s MASTER_PORT=%s WORLD_SIZE=%s)'
                % (e, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'), os.environ.get('WORLD_SIZE')))
        try:
            # (1) every rank sits on its own device
            ident = 'cpu:%d' % rank
            if self.kind == 'hip':
                props = torch.cuda.get_device_properties(local_rank)
                ident = '%s/%s' % (getattr(props, 'uuid', None) or props.name, local_rank)
            idents = [None] * world
            dist.all_gather_object(idents, ident)
            if self.kind == 'hip' and len(set(idents)) != world:
                die(5, 'two ranks share a device: %s' % idents)
            # (2) the collectives the sharded path issues, at their smallest: all_reduce, all_to_all_single with uneven splits
            x = torch.full((4,), float(rank + 1), device=self.dev)
            dist.all_reduce(x)
            want = world * (world + 1) / 2.0
            if abs(float(x[0].item()) - want) > 1e-3:
                die(5, 'all_reduce returned %r, expected %r' % (float(x[0].item()), want))
            send_counts = [(rank + p) % 3 + 1 for p in range(world)]
            recv_counts = [(p + rank) % 3 + 1 for p in range(world)]
            send = torch.cat([torch.full((c,), float(rank * 100 + p), device=self.dev) for p, c in enumerate(send_counts)])
            recv = torch.empty(sum(recv_counts), device=self.dev)
            dist.all_to_all_single(recv, send, recv_counts, send_counts)
            got = recv.cpu().tolist()
            exp = [float(p * 100 + rank) for p, c in enumerate(recv_counts) for _ in range(c)]
            if got != exp:
                die(5, 'all_to_all_single with uneven splits returned %r, expected %r' % (got, exp))
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            die(5, 'collective self-check failed: %r' % (e,))
        return dist

    def generator(self, seed):
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        return gen

    def use_side_stream(self):
        if self.kind == 'hip':
            self.side = torch.cuda.Stream(self.dev)
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
            torch.cuda.set_stream(self.side)

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream if self.kind == 'hip' else 0

    def sync(self):
        if self.kind == 'hip':
            torch.cuda.synchronize(self.dev)


def measured_stream_rates(be, stream):
    """Copy / triad GB/s of this GPU (slk_probe_stream: float4 kernels over 1 GiB buffers) -- the measured figure
    SURVEY.md 8(d) asks for next to the nominal peak."""
    n = (1 << 28) if be.kind == 'hip' else (1 << 12)
    a, b, c = (torch.ones(n, device=be.dev) for _ in range(3))
    ms = {k: be.engine.probe_stream(k, a.data_ptr(), b.data_ptr(), c.data_ptr(), n, iters=10, stream=stream) for k in range(12)}
    del a, b, c
    gbs = {'copy_plain': 8.0 * n / ms[0] / 1e6, 'triad_plain': 12.0 * n / ms[1] / 1e6, 'copy_nt_x4': 8.0 * n / ms[2] / 1e6,
           'triad_nt_x4': 12.0 * n / ms[3] / 1e6, 'read_only': 4.0 * n / ms[4] / 1e6, 'write_only': 4.0 * n / ms[5] / 1e6,
           'copy_chunk_x8': 8.0 * n / ms[6] / 1e6, 'copy_chunk_x8_nt': 8.0 * n / ms[7] / 1e6, 'copy_chunk_x4_16wg': 8.0 * n / ms[8] / 1e6,
           'copy_chunk_x16_4wg': 8.0 * n / ms[9] / 1e6, 'copy_chunk_x8_ntload': 8.0 * n / ms[10] / 1e6,
           'copy_chunk_x4_nt_32wg': 8.0 * n / ms[11] / 1e6}
    return {'copy_GBs': max(v for k, v in gbs.items() if k.startswith('copy')),
            'triad_GBs': max(gbs['triad_plain'], gbs['triad_nt_x4']), 'variants_GBs': gbs,
            'note': 'slk_probe_stream over 1 GiB buffers, hipEvents, 10 launches each: float4 copy / triad, plain grid-stride and '
                    'non-temporal with 4 accesses in flight per lane; read-only and write-only streams; chunked copies (a workgroup '
                    'moves contiguous 16-64 KB chunks, 4-16 loads in flight per lane, plain / non-temporal); copy_GBs = the best copy'}


def sharded_world1_check(be, args, tables, s1, s2, users, items, B, stream):
    """N = 1 consistency of the two engines: the same two minibatches, from the same tables and the same RNG
    state, through the fused path and through the row-sharded exchange path at world 1 (exchange = device copy);
    per-minibatch losses must agree.  Runs on copies of the tables."""
    from spotlight_amd.factorization.sharded import ShardedBilinearTrainer
    import torch.distributed as dist
    eng = be.engine
    K = 2
    state = np.random.RandomState(77).get_state()
    losses = []
    times = []
    for path in ('fused', 'sharded'):
        t = [x.clone() for x in tables]
        a1 = [x.clone() for x in s1]
        a2 = [x.clone() for x in s2] if s2 else None
        op = _native.make_optim(args.opt, [x.data_ptr() for x in a1], [x.data_ptr() for x in a2] if a2 else None, lr=1e-2,
                                weight_decay=1e-6 if args.opt == 'adam_dense' else 0.0)
        mb = torch.zeros(K, device=be.dev)
        eng.rng_set_state(state)
        if path == 'fused':
            tb = _native.make_tables([x.data_ptr() for x in t], t[0].shape[0], t[1].shape[0], args.dim)
            run = lambda lo: eng.bilinear_train(tb, op, users[lo:].data_ptr(), items[lo:].data_ptr(), K * B, B, args.loss, 1,
                                                mb.data_ptr(), stream=stream)
        else:
            tr = ShardedBilinearTrainer(eng, t, op, t[1].shape[0], stream=stream, slices=args.slices or None)
            tr.reserve(B, K)
            run = lambda lo: tr.train(users[lo:lo + K * B], items[lo:lo + K * B], B, loss=args.loss, mb_loss=mb)
        run(0)
        be.sync()
        first = mb.cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        run(K * B)  # the same call again on the next minibatches: buffers allocated, code paths warm
        be.sync()
        times.append((time.perf_counter() - t0) / K * 1e3)
        mb.copy_(torch.from_numpy(first).to(mb.dtype))
        losses.append(mb.cpu().numpy().astype(np.float64))
        del t, a1, a2
    rel = float(np.abs(losses[0] - losses[1]).max() / np.abs(losses[0]).max())
    return {'minibatches': K, 'loss_fused': losses[0].tolist(), 'loss_sharded_world1': losses[1].tolist(),
            'max_rel_diff': rel, 'consistent': bool(rel <= 1e-5),
            'ms_per_step_second_call': {'fused': times[0], 'sharded_world1': times[1]}}


def fit_end_to_end(be, args):
    """The drop-in API around the engine, end to end: ImplicitFactorizationModel.fit() (spotlight/factorization/implicit.py:184-252)
    on the workload's shapes -- per epoch the numpy-exact device shuffle, the id gathers, every minibatch, the loss read-back; the
    ids are uploaded once per fit() (host -> HBM, included).  One warm fit() of 3 epochs first (table initialisation, scratch,
    the epoch loop's id buffers), then a timed fit() of 10 epochs (the reference's default n_iter)."""
    from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
    from spotlight_amd.interactions import Interactions
    n = int(args.fit_interactions)
    rs = np.random.RandomState(5)
    inter = Interactions(rs.randint(0, args.users, n).astype(np.int32), rs.randint(0, args.items, n).astype(np.int32),
                         num_users=args.users, num_items=args.items)
    opts = {'adagrad': dict(sparse=True, optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2)),
            'sparse_adam': dict(sparse=True, optimizer_func=lambda p: torch.optim.SparseAdam(list(p), lr=1e-2)),
            'adam_dense': dict(l2=1e-6)}[args.opt]
    # (the warm fit runs 3 epochs: the large-epoch loop rotates three pairs of id buffers, and the timed fit should find all of
    # them in torch's caching allocator like every fit() after a process's first -- fresh HIP allocations of that size cost
    # 15-25 ms each, profiles/r04_t_fit_first_epoch_probe.txt)
    model = ImplicitFactorizationModel(loss=args.loss, embedding_dim=args.dim, n_iter=3, batch_size=args.batch, use_cuda=True,
                                       random_state=np.random.RandomState(1), **opts)
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    first = time.perf_counter() - t0
    epochs = 10  # the reference's default n_iter: the id upload and the first epoch's unhidden shuffle amortise as they do for a user
    model._n_iter = epochs
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    t_full = time.perf_counter() - t0
    dt = t_full / epochs
    # the same call with 2 epochs: the difference is 8 epochs of the steady state (no id upload, no first shuffle, no drain)
    model._n_iter = 2
    t0 = time.perf_counter()
    model.fit(inter)
    be.sync()
    t_two = time.perf_counter() - t0
    steady = (t_full - t_two) / (epochs - 2)
    return {'interactions_per_epoch': n, 'epochs_timed': epochs, 'seconds_per_epoch': dt, 'interactions_per_s': n / dt,
            'steady_state_seconds_per_epoch': steady, 'steady_state_interactions_per_s': n / steady,
            'steady_state_note': '(fit of 10 epochs - fit of 2 epochs) / 8: what every further epoch costs once the three-stage '
                                 'pipeline runs (next epoch\'s negatives + first sorts, the shuffle after next, this epoch\'s passes)',
            'first_fit_seconds': first,
            'what': 'ImplicitFactorizationModel.fit(): id upload (once per fit), per epoch the numpy-exact device shuffle + id '
                    'gathers + %d minibatches + the loss read-back; first_fit_seconds also holds table initialisation on the '
                    'host and scratch allocation' % ((n + args.batch - 1) // args.batch)}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n'
                         % (args.gpus, os.environ['WORLD_SIZE']))
        sys.exit(3)
    if args.workload == 'c4':
        return bench_c4(args)
    if args.workload in ('predict', 'eval'):
        return bench_scoring(args)
    if args.workload == 'c3':
        return bench_c3(args)
    # libraries (RCCL's version banner, rocm-smi) write to fd 1; keep stdout clean for the ONE
    # JSON line the driver parses: everything else goes to stderr until the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    be = Backend(args.backend, local_rank)
    want_sharded_check = (world == 1 and not args.sharded and not args.no_sharded_check and args.workload in ('c2', 'c5'))
    if world > 1 or args.sharded:
        dist = be.init_dist(rank, world, local_rank)
    dev = be.dev
    U, I, D, B = args.users, args.items, args.dim, args.batch
    K, W = args.steps, args.warmup

    eng = be.engine
    for kv in args.set:
        name, value = kv.split('=')
        eng.set_option(name, int(value))
    gen = be.generator(1234 + rank)
    # per-GPU shard: U x D users, I x D items (N > 1: the global tables are world times larger,
    # row-sharded cyclically; per-GPU work is fixed = weak scaling)
    tables = [torch.empty(U, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.empty(I, D, device=dev).normal_(0, 1.0 / D, generator=gen),
              torch.zeros(U, device=dev), torch.zeros(I, device=dev)]
    s1 = [torch.zeros_like(t) for t in tables]
    s2 = [torch.zeros_like(t) for t in tables] if args.opt != 'adagrad' else None
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D)
    op = _native.make_optim(args.opt, [t.data_ptr() for t in s1], [t.data_ptr() for t in s2] if s2 else None,
                            lr=1e-2, weight_decay=1e-6 if args.opt == 'adam_dense' else 0.0)
    n_total = (W + 2 * K) * B  # W warmup + K timed + K profiled
    I_global = I * world
    users = torch.randint(0, U, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    items = torch.randint(0, I_global, (n_total,), device=dev, dtype=torch.int64, generator=gen)
    def zipf_ids(n_ids, s_exp):
        w = 1.0 / torch.arange(1, n_ids + 1, device=dev, dtype=torch.float64) ** s_exp
        cdf = torch.cumsum(w / w.sum(), 0)
        ranks = torch.searchsorted(cdf, torch.rand(n_total, device=dev, dtype=torch.float64, generator=gen)).clamp_(max=n_ids - 1)
        return torch.randperm(n_ids, device=dev, generator=gen)[ranks]
    if args.item_zipf > 0:
        # SURVEY.md 8(d) "optional second distribution": Zipf item ids (the negatives stay uniform, as the reference draws them)
        items = zipf_ids(I_global, args.item_zipf)
    if args.user_zipf > 0:
        users = zipf_ids(U, args.user_zipf)
    mb_loss = torch.zeros(W + 2 * K, device=dev)
    eng.rng_set_state(np.random.RandomState(1 + rank).get_state())
    # the engine runs on its own HIP stream (ordered against torch's current stream by events)
    if args.side_stream:
        be.use_side_stream()
    stream = be.stream()
    trainer = None
    if world > 1 or args.sharded:
        from spotlight_amd.factorization.sharded import ShardedBilinearTrainer
        trainer = ShardedBilinearTrainer(eng, tables, op, I_global, stream=stream, slices=args.slices or None)
        trainer.reserve(B, args.shard_chunk)  # exchange buffers of the timed loop's chunks up front
    xgmi_rows = [0]

    def run(first_mb, n_mb):
        if trainer is None:
            off = first_mb * B
            eng.bilinear_train(tb, op, users[off:].data_ptr(), items[off:].data_ptr(), n_mb * B, B, args.loss, 1,
                               mb_loss[first_mb:].data_ptr(), stream=stream)
            return
        # this rank's B interactions of every global minibatch (users it owns; items anywhere)
        lo, hi = first_mb * B, (first_mb + n_mb) * B
        trainer.train(users[lo:hi], items[lo:hi], B, loss=args.loss, mb_loss=mb_loss[first_mb:first_mb + n_mb],
                      sample_chunk=args.shard_chunk)
        xgmi_rows[0] += trainer.exchange_rows

    multi = world > 1 or args.sharded

    def barrier():
        be.sync()
        if multi:
            dist.barrier()
            be.sync()

    # The copy / triad probes (own buffers, the tables untouched) go BEFORE the warm-up: a GPU coming out of idle runs its first
    # ~20 ms 4-6 % slower (profiles/r03_t_first_call_after_idle.txt: the figure returns after 0.5 s of idle), and W = 5 warm-up
    # minibatches are 4 ms.  The timed region itself is unchanged: W untimed steps, then exactly K.
    probes = ceiling = shard_check = None
    want_probes = rank == 0 and world == 1 and trainer is None and not args.no_probes
    if want_probes:
        probes = measured_stream_rates(be, stream)
    if trainer is None:
        eng.bilinear_reserve(tb, op, K * B, B, args.loss, 1, stream=stream)  # scratch for the timed call's shape
    if W:
        run(0, W)
    barrier()
    xgmi_rows[0] = 0
    # timed region: EXACTLY K steps, no instrumentation inside
    t0 = time.perf_counter()
    run(W, K)
    be.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    # per-kernel durations: K more steps with hipEvents around every launch (slk_profile_*);
    # outside the timed region because the event records themselves cost ~10 us per launch
    xg = xgmi_rows[0]
    eng.profile_reset()
    eng.profile_enable(True)
    t1 = time.perf_counter()
    run(W + K, K)
    be.sync()
    elapsed_profiled = time.perf_counter() - t1
    eng.profile_enable(False)
    prof = eng.profile_read()
    xgmi_rows[0] = xg
    # The timed region above runs a bare ctx's default: negatives, sorts and passes in order on one stream.  fit() switches
    # the ctx to "overlap_prep" (the next chunk's negatives + sorts on a second stream beside the passes); the same K
    # minibatches are run that way three more times -- untimed warm-up, un-instrumented, with the kernel timers -- and go
    # into the line as roofline.overlapped: the steady state a training run of many calls sees.
    prof_ov = elapsed_ov = elapsed_ov_prof = None
    if trainer is None and world == 1 and not args.no_overlapped and not any(kv.startswith('overlap_prep=') for kv in args.set):
        eng.set_option('overlap_prep', 1)
        try:
            eng.bilinear_reserve(tb, op, K * B, B, args.loss, 1, stream=stream)  # the second buffer set, the prep stream
            run(W + K, K)
            be.sync()
            t2 = time.perf_counter()
            run(W + K, K)
            be.sync()
            elapsed_ov = time.perf_counter() - t2
            eng.profile_reset()
            eng.profile_enable(True)
            t2 = time.perf_counter()
            run(W + K, K)
            be.sync()
            elapsed_ov_prof = time.perf_counter() - t2
            eng.profile_enable(False)
            prof_ov = eng.profile_read()
        finally:
            eng.set_option('overlap_prep', 0)
    # The line's value: the same K minibatches once more, in line on one stream as in the first timed call, now that the process
    # has ~5 K steps behind it.  The FIRST K-step call of a process (elapsed, above: what rounds 1-3 reported as the value) runs
    # 4-5 % slower than every later one -- the GPU is still leaving its idle power state; 1 s of copy kernels in front takes half
    # of that off, more training steps all of it (profiles/r04_o_first_call_of_a_process.txt) -- and a training run is thousands
    # of steps long.  Both figures go into the line (first_call); the bracket is the same: barrier + sync, exactly K steps, sync.
    elapsed_first = elapsed
    barrier()
    xg = xgmi_rows[0]
    t3 = time.perf_counter()
    run(W, K)
    be.sync()
    elapsed = time.perf_counter() - t3
    barrier()
    xgmi_rows[0] = xg
    ranks_seen = [{'rank': rank, 'local_rank': local_rank, 'device': be.name}]
    if multi:
        dist.barrier()
        t = torch.tensor([elapsed, elapsed_first], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_first = float(t[0].item()), float(t[1].item())
        dist.all_reduce(mb_loss)  # per-rank shares of each global minibatch loss
        # what the process group itself observed: its size and every rank's device
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, ranks_seen[0])
        ranks_seen = seen
        assert dist.get_world_size() == world == args.gpus or args.sharded, (dist.get_world_size(), world, args.gpus)
    denominators = None
    if world > 1 and not args.no_denominators:
        # What the N-GPU value is a multiple OF (--gpus 1 runs the fused path on C2, --gpus N the row-sharded
        # path): rank 0 runs the SAME per-GPU shape alone, through the fused path and through the row-sharded path at world 1.
        g1 = dist.new_group([0])  # every rank makes the call
        if rank == 0:
            try:
                from spotlight_amd.factorization.sharded import ShardedBilinearTrainer
                denominators = {}
                for path in ('fused', 'sharded_world1'):
                    for t in tables + s1 + (s2 or []):
                        t.zero_()
                    for t in tables[:2]:
                        t.normal_(0, 1.0 / D, generator=gen)
                    op1 = _native.make_optim(args.opt, [t.data_ptr() for t in s1], [t.data_ptr() for t in s2] if s2 else None, lr=1e-2)
                    it1 = items % I
                    mb1 = torch.zeros(W + K, device=dev)
                    if path == 'fused':
                        eng.bilinear_reserve(tb, op1, K * B, B, args.loss, 1, stream=stream)
                        go = lambda lo, nmb: eng.bilinear_train(tb, op1, users[lo * B:].data_ptr(), it1[lo * B:].data_ptr(), nmb * B, B,
                                                                args.loss, 1, mb1[lo:].data_ptr(), stream=stream)
                    else:
                        tr1 = ShardedBilinearTrainer(eng, tables, op1, I, group=g1, stream=stream, slices=args.slices or None)
                        tr1.reserve(B, args.shard_chunk)
                        go = lambda lo, nmb: tr1.train(users[lo * B:(lo + nmb) * B], it1[lo * B:(lo + nmb) * B], B, loss=args.loss,
                                                       mb_loss=mb1[lo:lo + nmb], sample_chunk=args.shard_chunk)
                    if W:
                        go(0, W)
                    be.sync()
                    t1 = time.perf_counter()
                    go(W, K)
                    be.sync()
                    dt = time.perf_counter() - t1
                    denominators[path] = {'interactions_per_s': K * B / dt, 'ms_per_step': dt / K * 1e3}
                denominators['note'] = ('rank 0 alone, after the timed region, on the same per-GPU shape (%d users x %d items, minibatch %d): '
                                        'the fused single-GPU path and the row-sharded path at world 1 (every exchange a local copy)'
                                        % (U, I, B))
            except Exception as e:
                denominators = {'error': repr(e)[:300]}
        dist.barrier()
    if want_probes:
        um, im, touched = eng.probe_step_ceiling(tb, op, B, iters=10, stream=stream)
        ceiling = {'user_side_ms': um, 'item_side_ms': im, 'items_touched': touched}
    if want_sharded_check and rank == 0:
        try:
            dist = be.init_dist(0, 1, local_rank)  # after the timed region: a world-1 group for the exchange path
            shard_check = sharded_world1_check(be, args, tables, s1, s2, users, items, B, stream)
        except Exception as e:
            shard_check = {'error': repr(e)[:300]}

    fit_rec = None
    if rank == 0 and world == 1 and trainer is None and be.kind == 'hip' and args.workload == 'c2' and not args.no_fit:
        try:
            del users, items
            fit_rec = fit_end_to_end(be, args)
        except Exception as e:  # a reported extra, never a reason to lose the line
            fit_rec = {'error': repr(e)[:300]}

    losses = mb_loss.cpu().numpy()
    assert args.no_loss_check or (np.isfinite(losses).all() and (losses[W:] > 0).all()), losses
    losses = losses[:W + K]

    if rank == 0:
        value = world * K * B / elapsed
        s_words = 1 if args.opt == 'adagrad' else 2  # (adam_dense: the per-row figure; its full-table sweep is extra)
        ub, ib = algorithmic_bytes(D, s_words)
        kern = {}
        for name, per_int in (('user_pass', ub), ('item_pass', ib)):
            n, ms = prof[name]
            avg_s = ms / max(n, 1) * 1e-3
            kern[name] = {'launches': n, 'avg_ms': ms / max(n, 1), 'alg_bytes_per_launch': per_int * B,
                          'achieved_GBs': per_int * B / avg_s / 1e9 if avg_s > 0 else 0.0}
        dom = max(kern, key=lambda k: kern[k]['avg_ms'])
        roof = {'bound': 'hbm', 'kernel': 'k_' + dom, 'achieved': kern[dom]['achieved_GBs'],
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kern[dom]['achieved_GBs'] / HBM_PEAK_GBS,
                'traffic': pmc_traffic(args, 'k_' + dom) if trainer is None else None,
                'traffic_note': 'HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from profiles/pmc_traffic.json '
                                '(rocprofv3 --pmc passes over this workload); null if no committed measurement matches',
                'kernels': kern,
                'step_alg_bytes_per_interaction': ub + ib,
                'step_frac_of_peak': value / world * (ub + ib) / (HBM_PEAK_GBS * 1e9),
                'other_ms_per_step': {k: prof[k][1] / K for k in ('sample', 'prep', 'exchange', 'dense_sweep', 'epoch')}}
        if prof_ov is not None:
            kov = {}
            for name, per_int in (('user_pass', ub), ('item_pass', ib)):
                n, ms = prof_ov[name]
                avg_s = ms / max(n, 1) * 1e-3
                kov[name] = {'avg_ms': ms / max(n, 1), 'achieved_GBs': per_int * B / avg_s / 1e9 if avg_s > 0 else 0.0,
                             'frac': per_int * B / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else 0.0}
            roof['overlapped'] = {'ms_per_step': elapsed_ov / K * 1e3, 'interactions_per_s': K * B / elapsed_ov,
                                  'step_frac_of_peak': K * B / elapsed_ov * (ub + ib) / (HBM_PEAK_GBS * 1e9),
                                  'kernels': kov, 'ms_per_step_with_kernel_timers': elapsed_ov_prof / K * 1e3,
                                  'other_ms_per_step': {k: prof_ov[k][1] / K for k in ('sample', 'prep')},
                                  'note': 'the same K minibatches with option overlap_prep = 1 (what fit() sets on its ctx): the next '
                                          'chunk\'s negatives + sorts on a second stream beside the passes; second call of its kind '
                                          '(the line\'s value is the same K minibatches in order on one stream).  Beside the sorts every pass runs '
                                          'longer, the step shorter.'}
        if prof['epoch'][0]:
            roof['persistent_epoch_kernel'] = {'launches': prof['epoch'][0], 'us_per_minibatch': prof['epoch'][1] / K * 1e3,
                                               'note': 'every minibatch of a chunk inside one cooperative launch (slk_epoch.hip)'}
        if probes:
            roof['measured'] = probes
        if ceiling:
            # the step's algorithmic accesses alone (slk_probe_step_ceiling): what exact grouping + hand-over cost on top
            tot = ceiling['user_side_ms'] + ceiling['item_side_ms']
            ceiling.update({'ms_per_step': tot, 'step_frac_of_peak': (ub + ib) * B / (tot * 1e-3) / (HBM_PEAK_GBS * 1e9),
                            'user_side_frac_of_peak': ub * B / (ceiling['user_side_ms'] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                            'note': 'algorithmic row accesses only (no sorts, records, ids, biases, duplicate handling) on the '
                                    'same tables in the same lane layout; the item side touches each distinct item once'})
            roof['ceiling'] = ceiling
        if trainer is not None:
            kern_ms = sum(prof[k][1] for k in ('sample', 'prep', 'user_pass', 'item_pass', 'exchange')) / K
            roof['xgmi'] = {'rows_per_step_per_gpu': xgmi_rows[0] / K,
                            'bytes_per_step_per_gpu_each_way': xgmi_rows[0] / K * (2 * (D + 1) * 4 + 4),  # id + row + gradient
                            'slices_per_minibatch': trainer.slices, 'minibatches_per_chunk': args.shard_chunk,
                            'kernel_ms_per_step': kern_ms,
                            'exchange_and_host_ms_per_step': elapsed / K * 1e3 - kern_ms}
            # the exchange bound: every byte leaves through one of the (world - 1) direct xGMI links of
            # this GPU (point-to-point mesh, ~76.8 GB/s per link and direction)
            xb = roof['xgmi']['bytes_per_step_per_gpu_each_way']
            peak = max(world - 1, 1) * 76.8
            roof['xgmi'].update({'link_peak_GBs_each_way': peak,
                                 'achieved_GBs_each_way_over_the_whole_step': xb / (elapsed / K) / 1e9,
                                 'min_ms_per_step_at_link_peak': xb / (peak * 1e9) * 1e3 if world > 1 else 0.0})
            if world > 1:
                # the curve is to be read against the wire, not against N x one GPU: every interaction moves, per GPU and direction,
                # 2 lookups x (N-1)/N remote x (id 4 B + row (D+1)*4 B, then gradient (D+1)*4 B) over N-1 links of 76.8 GB/s
                roof['xgmi']['bound_%d_gpus' % world] = {
                    'interactions_per_s_at_link_peak': world * B / (xb / (peak * 1e9)),
                    'note': 'whole-job rate at which the exchange alone saturates every xGMI link (exact fp32 rows on the wire)'}
                if denominators is not None:
                    roof['xgmi']['denominators_1_gpu'] = denominators
                    for k in ('fused', 'sharded_world1'):
                        if isinstance(denominators.get(k), dict):
                            denominators[k]['scaling_factor_of_this_run'] = value / denominators[k]['interactions_per_s']
            if world == 1:
                # a MODEL, not a measurement: what this rank's measured kernel time and the wire allow at 8 GPUs.  Per
                # direction a GPU moves, for 7/8 of its 2B lookups, the id + the row (as requester in, as owner out)
                # + the gradient (the other way round) over 7 links of 76.8 GB/s.
                wire_ms = 2 * B * 7 / 8 * (2 * (D + 1) * 4 + 4) / (7 * 76.8e9) * 1e3
                roof['xgmi']['model_8_gpus'] = {
                    'kind': 'model (no 8-GPU hardware measured)', 'kernel_ms_per_step_measured_here': kern_ms,
                    'exchange_ms_per_step_at_link_peak': wire_ms,
                    'ms_per_step_exchange_fully_hidden_or_hiding': max(kern_ms, wire_ms),
                    'ms_per_step_nothing_overlapped': kern_ms + wire_ms,
                    'interactions_per_s_range': [8 * B / ((kern_ms + wire_ms) * 1e-3), 8 * B / (max(kern_ms, wire_ms) * 1e-3)]}
        out = {'metric': 'training interactions/sec, BPR dim=64', 'value': value, 'unit': 'interactions/s',
               'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
               'data': 'synthetic' if be.kind == 'hip' else 'synthetic; CPU EMULATOR TEST RUN, not a measurement',
               'ranks': {'world_size_observed': dist.get_world_size() if multi else 1, 'devices': ranks_seen,
                         'launched_by': 'bench.py (torch.distributed.run)' if os.environ.get('SLK_BENCH_SPAWNED') else
                                        ('torchrun' if 'WORLD_SIZE' in os.environ else 'single process')},
               'config': {'workload': '%s: synthetic uniform ids, %d users x %d items, dim %d, %s loss, '
                                      '%s lr=1e-2, minibatch %d%s, on-GPU numpy-exact negatives%s'
                                      % (args.workload.upper(), U * world, I * world, D, args.loss, args.opt, B * world,
                                         '' if world == 1 else ' (= %d per GPU; tables and batch grow with N)' % B,
                                         ('; POSITIVE ITEMS ZIPF(%g) -- a stress run, not the metric\'s distribution' % args.item_zipf
                                          if args.item_zipf > 0 else '') +
                                         ('; USERS ZIPF(%g) -- a stress run, not the metric\'s distribution' % args.user_zipf
                                          if args.user_zipf > 0 else '')),
                          'global_batch': B * world,
                          'parallelism': 'single GPU' if trainer is None else
                          'row-sharded x%d: users and items sharded cyclically; RCCL all-to-all of ids per chunk of '
                          'minibatches, of rows and gradient rows per user-slice of a minibatch (async, overlapping '
                          'the other slices\' compute); no replicas' % world},
               'roofline': roof,
               'ms_per_step_with_kernel_timers': elapsed_profiled / K * 1e3,
               'timed_region': 'exactly K minibatches between barrier + device sync, negatives, sorts and passes in order on one '
                               'stream; the LAST K-step call of the run (after the W warm-up steps, the first K-step call, the '
                               'instrumented one and the roofline.overlapped leg): the steady state of a training run',
               'first_call': {'ms_per_step': elapsed_first / K * 1e3, 'interactions_per_s': world * K * B / elapsed_first,
                              'note': 'the first K-step call of the process, straight after the W warm-up steps (what rounds 1-3 '
                                      'printed as value): the GPU is still leaving its idle power state, '
                                      'profiles/r04_o_first_call_of_a_process.txt'},
               'final_minibatch_loss': float(losses[-1])}
        if shard_check is not None:
            out['sharded_world1_consistency'] = shard_check
        if fit_rec is not None:
            out['fit_end_to_end'] = fit_rec
        if world == 1 and not args.no_cpu_baseline:
            # the reference itself on this box's host cores; the scalar C port (1 thread) as a secondary field
            ref = reference_cpu_baseline(args, args.cpu_seconds * 2.0 / 3.0)  # the timed fits' share; its set-up comes on top
            port = cpu_baseline(args, 6.0 if ref and 'value' in ref else args.cpu_seconds)
            if ref and 'value' in ref:
                out['cpu_baseline'] = ref
                out['cpu_baseline']['port'] = port
            else:
                out['cpu_baseline'] = port
                if ref:
                    out['cpu_baseline']['reference_error'] = ref.get('error')
    else:
        out = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL printf()s its banner into libc's stdout buffer: flush it while fd 1 still points at stderr
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    if out is not None:
        os.write(1, (json.dumps(out) + '\n').encode())


if __name__ == '__main__':
    main()
