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
from benchlib.cpu import cpu_baseline, reference_cpu_baseline  # noqa: E402
from benchlib.dist import Backend, spawn_ranks  # noqa: E402
from benchlib.probes import fit_end_to_end, measured_stream_rates, sharded_world1_check  # noqa: E402
from benchlib.report import build_roofline  # noqa: E402


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
    ap.add_argument('--workload', default=None, choices=['c1', 'c2', 'c3', 'c4', 'c5', 'predict', 'eval'],
                    help='c1: the reference\'s own test configuration (MovieLens-100K shape, dim 32, bpr, default Adam, minibatch '
                         '1024), the whole 10-epoch fit(); '
                         'c2: BilinearNet BPR step (the headline metric; the default at --gpus 1); c3: adaptive hinge n=5 over a '
                         'BloomEmbedding item table, dim 128; c4: PoolNet sequence step; c5: the per-GPU shard of the 1B-item x '
                         '100M-user table (12.5M users x 125M items per GPU; at --gpus 8 the full C5; the default at --gpus > 1: '
                         'the row-sharded configuration BASELINE.json names).  Explicit --users / --items override the shape.')
    ap.add_argument('--no-denominators', action='store_true',
                    help='N > 1: skip rank 0\'s world-1 runs of the same per-GPU shape (fused path, row-sharded path) that give '
                         'the scaling factor its stated denominators')
    ap.add_argument('--seq-len', type=int, default=200)
    ap.add_argument('--pad-frac', type=float, default=0.0,
                    help='c4: this share of all positions is left padding (SURVEY.md 8(d): the 20 %%-left-padded variant = 0.2)')
    ap.add_argument('--n-neg', type=int, default=5,
                    help='--loss adaptive_hinge: num_negative_samples (5: the reference\'s default, factorization/implicit.py:88)')
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
    ap.add_argument('--no-bias-shadow', action='store_true',
                    help='item tables of 2^24 rows and more: keep the item biases and their Adagrad accumulator in two arrays '
                         '(default: interleaved for the run, as fit() trains such tables -- slk_bias_shadow_begin, opened before '
                         'the warm-up and closed after the last timed call)')
    ap.add_argument('--no-user-pingpong', action='store_true',
                    help='fused path, pair losses, row-sparse optimizers, minibatches of 2^17 and more: keep the user table ONCE '
                         '(default: doubled for the run, as fit() trains such minibatches -- slk_user_pingpong_begin: the user pass '
                         'writes updated rows to the other copy and no pre-step-row record; opened before the warm-up, closed '
                         'after the last timed call, its closing merge timed separately as roofline.pingpong_merge_ms)')
    ap.add_argument('--user-pingpong-min-batch', type=int, default=1 << 17,
                    help='minibatch size from which the run trains on the doubled user table (tests force it at small sizes)')
    ap.add_argument('--bias-shadow-min-items', type=int, default=1 << 24,
                    help='item rows per GPU from which the run trains on the bias shadow (tests force it at small sizes)')
    ap.add_argument('--no-loss-check', action='store_true', help='measurement of debug modes whose results are meaningless')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true',
                    help='N=1, C2: skip the compact legs of the other BASELINE.json configurations (the `configs` object of the line)')
    ap.add_argument('--cpu-seconds', type=float, default=30.0)
    args = ap.parse_args()
    custom = args.users is not None or args.items is not None
    if args.workload is None:
        args.workload = 'c5' if (args.gpus > 1 and not custom) else 'c2'
    shape = {'c5': (12_500_000, 125_000_000)}.get(args.workload, (10_000_000, 1_000_000))
    args.users = args.users if args.users is not None else shape[0]
    args.items = args.items if args.items is not None else shape[1]
    return args


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n'
                         % (args.gpus, os.environ['WORLD_SIZE']))
        sys.exit(3)
    if args.workload in ('c1', 'c3', 'c4', 'predict', 'eval'):
        # diagnostic workloads (not the headline metric): each returns its own record, printed as one JSON line
        if args.workload == 'c1':
            from benchlib.c1 import bench_c1 as leg
        elif args.workload == 'c4':
            from benchlib.c4 import bench_c4 as leg
        elif args.workload == 'c3':
            from benchlib.c3 import bench_c3 as leg
        else:
            from benchlib.scoring import bench_scoring as leg
        print(json.dumps(leg(args)), flush=True)
        return
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
    want_sharded_check = (world == 1 and not args.sharded and not args.no_sharded_check and args.workload in ('c2', 'c5')
                          and args.loss != 'adaptive_hinge')
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
    # (user biases: zeros, as the reference initialises them -- and bpr / hinge never move them; checked on the device, as fit() does,
    # before the hint is given: include/spotlight_hip.h, SLK_TABLES_USER_BIAS_ZERO)
    tb = _native.make_tables([t.data_ptr() for t in tables], U, I, D, user_bias_zero=not bool(tables[2].any()))
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
        trainer = ShardedBilinearTrainer(eng, tables, op, I_global, stream=stream, slices=args.slices or None,
                                         user_bias_zero=not bool(tables[2].any()))
        trainer.reserve(B, args.shard_chunk)  # exchange buffers of the timed loop's chunks up front
    xgmi_rows = [0]
    xgmi_bytes = [0, 0]  # measured by the trainer: bytes handed to the collectives for other ranks (with / without slot padding)
    # fit() trains item tables this large with {bias, Adagrad accumulator} interleaved (factorization/implicit.py:
    # _BIAS_SHADOW_MIN_ITEMS); the scope opens here, outside every timed region, and closes after the last one
    bias_shadowed = (args.opt == 'adagrad' and I >= args.bias_shadow_min_items and (B >= 4096 or trainer is not None)
                     and not args.no_bias_shadow)
    shadow_scope = (eng.bias_shadow(tb, op, stream=stream, enabled=bias_shadowed) if trainer is None else
                    trainer.bias_shadow(enabled=bias_shadowed))  # (row-sharded: the owner-side gather and item pass index it)
    shadow_scope.__enter__()
    # ... and minibatches this large on a doubled user table (factorization/implicit.py: _USER_PINGPONG_MIN_BATCH): no pre-step-row
    # record is written; the scope's closing merge (rows whose current copy is the ctx's -> the caller's table) is timed below
    pingponged = (trainer is None and args.loss in ('bpr', 'hinge', 'pointwise') and args.opt in ('adagrad', 'sparse_adam')
                  and B >= args.user_pingpong_min_batch and not args.no_user_pingpong)
    pp_scope = eng.user_pingpong(tb, op, stream=stream, enabled=pingponged)
    pp_scope.__enter__()
    pingponged = pp_scope.active  # (SLK_ENOMEM: the run goes on in the one-table layout)

    n_neg = args.n_neg if args.loss == 'adaptive_hinge' else 1

    def run(first_mb, n_mb):
        if trainer is None:
            off = first_mb * B
            eng.bilinear_train(tb, op, users[off:].data_ptr(), items[off:].data_ptr(), n_mb * B, B, args.loss, n_neg,
                               mb_loss[first_mb:].data_ptr(), stream=stream)
            return
        # this rank's B interactions of every global minibatch (users it owns; items anywhere)
        lo, hi = first_mb * B, (first_mb + n_mb) * B
        trainer.train(users[lo:hi], items[lo:hi], B, loss=args.loss, mb_loss=mb_loss[first_mb:first_mb + n_mb],
                      sample_chunk=args.shard_chunk, n_neg=args.n_neg if args.loss == 'adaptive_hinge' else None)
        xgmi_rows[0] += trainer.exchange_rows
        xgmi_bytes[0] += trainer.exchange_bytes
        xgmi_bytes[1] += trainer.exchange_payload_bytes

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
        eng.bilinear_reserve(tb, op, K * B, B, args.loss, n_neg, stream=stream)  # scratch for the timed call's shape
    if W:
        run(0, W)
    barrier()
    xgmi_rows[0] = 0
    xgmi_bytes[:] = [0, 0]
    # timed region: EXACTLY K steps, no instrumentation inside
    t0 = time.perf_counter()
    run(W, K)
    be.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    # per-kernel durations: K more steps with hipEvents around every launch (slk_profile_*);
    # outside the timed region because the event records themselves cost ~10 us per launch
    xg, xgb = xgmi_rows[0], list(xgmi_bytes)
    eng.profile_reset()
    eng.profile_enable(True)
    single0 = eng.get_stat('single_minibatches')
    t1 = time.perf_counter()
    run(W + K, K)
    be.sync()
    elapsed_profiled = time.perf_counter() - t1
    eng.profile_enable(False)
    prof = eng.profile_read()
    # minibatches of the instrumented call whose once-only items were updated by the USER pass (option "item_single_min_items":
    # catalogues far larger than a minibatch): the pass then carries those items' algorithmic bytes too
    single_mb = eng.get_stat('single_minibatches') - single0
    xgmi_rows[0] = xg
    xgmi_bytes[:] = xgb
    # The timed region above runs a bare ctx's default: negatives, sorts and passes in order on one stream.  fit() switches
    # the ctx to "overlap_prep" (the next chunk's negatives + sorts on a second stream beside the passes); the same K
    # minibatches are run that way three more times -- untimed warm-up, un-instrumented, with the kernel timers -- and go
    # into the line as roofline.overlapped: the steady state a training run of many calls sees.
    prof_ov = elapsed_ov = elapsed_ov_prof = None
    if trainer is None and world == 1 and not args.no_overlapped and not any(kv.startswith('overlap_prep=') for kv in args.set):
        eng.set_option('overlap_prep', 1)
        try:
            eng.bilinear_reserve(tb, op, K * B, B, args.loss, n_neg, stream=stream)  # the second buffer set, the prep stream
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
    xg, xgb = xgmi_rows[0], list(xgmi_bytes)
    t3 = time.perf_counter()
    run(W, K)
    be.sync()
    elapsed = time.perf_counter() - t3
    barrier()
    xgmi_rows[0] = xg
    xgmi_bytes[:] = xgb
    be.sync()
    t4 = time.perf_counter()
    pp_scope.__exit__(None, None, None)  # the user table is whole again (the probes and checks below read it)
    be.sync()
    pingpong_merge_ms = (time.perf_counter() - t4) * 1e3 if pingponged else None
    shadow_scope.__exit__(None, None, None)  # (the probes and checks below read the two arrays)
    ranks_seen = [{'rank': rank, 'local_rank': local_rank, 'device': be.name}]
    if multi:
        ranks_seen[0]['exchange_rows_timed_call'] = int(xgmi_rows[0])  # lookups of this rank that crossed to another rank, K steps
        # MEASURED: the bytes this rank handed to the collectives for other ranks in those K steps (ids + rows + gradient rows;
        # slot padding included / excluded) -- to be read against roofline.xgmi's modelled bytes_per_step_per_gpu_each_way
        ranks_seen[0]['exchange_bytes_timed_call'] = int(xgmi_bytes[0])
        ranks_seen[0]['exchange_payload_bytes_timed_call'] = int(xgmi_bytes[1])
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
                    # both paths start from the SAME tables and the same negative stream: their per-minibatch losses must agree
                    # (tests/test_bench_cli.py) -- the two rates are then rates of the same work
                    for t in tables + s1 + (s2 or []):
                        t.zero_()
                    gen.manual_seed(4321)
                    for t in tables[:2]:
                        t.normal_(0, 1.0 / D, generator=gen)
                    eng.rng_set_state(np.random.RandomState(4321).get_state())
                    op1 = _native.make_optim(args.opt, [t.data_ptr() for t in s1], [t.data_ptr() for t in s2] if s2 else None, lr=1e-2)
                    it1 = items % I
                    mb1 = torch.zeros(W + K, device=dev)
                    if path == 'fused':
                        eng.bilinear_reserve(tb, op1, K * B, B, args.loss, n_neg, stream=stream)
                        go = lambda lo, nmb: eng.bilinear_train(tb, op1, users[lo * B:].data_ptr(), it1[lo * B:].data_ptr(), nmb * B, B,
                                                                args.loss, n_neg, mb1[lo:].data_ptr(), stream=stream)
                        scope = eng.bias_shadow(tb, op1, stream=stream, enabled=bias_shadowed and B >= 4096)
                    else:
                        tr1 = ShardedBilinearTrainer(eng, tables, op1, I, group=g1, stream=stream, slices=args.slices or None,
                                                     user_bias_zero=not bool(tables[2].any()))
                        tr1.reserve(B, args.shard_chunk)
                        go = lambda lo, nmb: tr1.train(users[lo * B:(lo + nmb) * B], it1[lo * B:(lo + nmb) * B], B, loss=args.loss,
                                                       mb_loss=mb1[lo:lo + nmb], sample_chunk=args.shard_chunk)
                        scope = tr1.bias_shadow(enabled=bias_shadowed)
                    with scope:  # the item-bias layout of the N-GPU run itself
                        if W:
                            go(0, W)
                        be.sync()
                        t1 = time.perf_counter()
                        go(W, K)
                        be.sync()
                        dt = time.perf_counter() - t1
                    denominators[path] = {'interactions_per_s': K * B / dt, 'ms_per_step': dt / K * 1e3,
                                          'minibatch_losses': [float(x) for x in mb1.cpu().numpy()]}
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

    legs = None
    default_shape = (args.users, args.items, args.dim, args.batch, args.loss, args.opt) == (10_000_000, 1_000_000, 64, 1 << 20, 'bpr', 'adagrad')
    if (rank == 0 and world == 1 and trainer is None and be.kind == 'hip' and args.workload == 'c2' and default_shape
            and not args.no_configs and not args.item_zipf and not args.user_zipf):
        from benchlib.legs import run_config_legs
        try:
            legs = run_config_legs(sum((['--set', kv] for kv in args.set), []))
        except Exception as e:  # noqa: BLE001
            legs = {'error': repr(e)[:300]}

    losses = mb_loss.cpu().numpy()
    assert args.no_loss_check or (np.isfinite(losses).all() and (losses[W:] > 0).all()), losses
    losses = losses[:W + K]

    if rank == 0:
        value, roof = build_roofline(args, world=world, K=K, B=B, D=D, elapsed=elapsed, prof=prof, prof_ov=prof_ov,
                                     elapsed_ov=elapsed_ov, elapsed_ov_prof=elapsed_ov_prof, probes=probes, ceiling=ceiling,
                                     trainer=trainer, xgmi_rows=xgmi_rows, denominators=denominators, xgmi_bytes=xgmi_bytes)
        if pingpong_merge_ms is not None:
            roof['pingpong_merge_ms'] = pingpong_merge_ms
        if single_mb == K and trainer is None:
            # the single-occurrence fast path ran in every minibatch: the two kernels' shares of the step's algorithmic bytes are no
            # longer 1576 / 1560 -- the user pass updates the items that occur once (nearly all on such a catalogue), the item pass
            # only walks the sorted list.  The dominant kernel is priced on the WHOLE step's algorithmic bytes (an upper bound of its
            # share: the few multi-occurrence items are still the item pass's)
            ku = roof['kernels']['user_pass']
            step_b = roof['step_alg_bytes_per_interaction'] * B
            ku['alg_bytes_per_launch'] = step_b
            ku['achieved_GBs'] = step_b / (ku['avg_ms'] * 1e-3) / 1e9
            roof['kernels']['item_pass']['alg_bytes_per_launch'] = 0
            roof['kernels']['item_pass']['achieved_GBs'] = 0.0
            roof.update({'kernel': 'k_user_pass', 'achieved': ku['achieved_GBs'], 'frac': ku['achieved_GBs'] / roof['peak'], 'traffic': None,
                         'single_occurrence_fast_path': 'every minibatch: items that occur once in their minibatch are updated by the user '
                                                        'pass (include/spotlight_hip.h, option item_single_min_items); the user pass is priced '
                                                        'on the whole step\'s algorithmic bytes, the item pass on none'})
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
                          'item_bias_layout': ('{bias, Adagrad accumulator} interleaved for the run (slk_bias_shadow_begin before the '
                                               'warm-up, _end after the last timed call: what fit() does on item tables this large)'
                                               if bias_shadowed else 'two arrays (torch layout)'),
                          'user_row_layout': ('the user table doubled for the run (slk_user_pingpong_begin before the warm-up, _end after '
                                              'the last timed call: what fit() does for minibatches this large) -- updated rows go to '
                                              'the other copy, no pre-step-row record is written; the closing merge is '
                                              'roofline.pingpong_merge_ms' if pingponged else 'one table, pre-step rows recorded per position'),
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
        if legs is not None and isinstance(out.get('cpu_baseline'), dict):
            # the reference's own runs of two legs' workloads, timed by the cpu_baseline leg (the one place that runs the staged
            # reference): its CPU fit() of the C1 call beside configs.c1, and the reference on this GPU through stock PyTorch-ROCm
            cb = out['cpu_baseline']
            ref_c1 = cb.get('c1_reference_fit')
            if isinstance(legs.get('c1'), dict) and isinstance(ref_c1, dict) and 'fit_seconds' in ref_c1:
                legs['c1']['reference_cpu_fit'] = {'fit_seconds': round(ref_c1['fit_seconds'], 4), 'threads': ref_c1['threads'],
                                                   'interactions_per_s': float('%.5g' % ref_c1['interactions_per_s'])}
            hip = cb.get('reference_on_hip')
            if isinstance(hip, dict):
                legs['reference_on_hip'] = ({'ms_per_step': round(hip['ms_per_minibatch'], 3), 'value': float('%.5g' % hip['interactions_per_s']),
                                             'unit': 'interactions/s', 'steps': hip['minibatches_per_fit'], 'warmup': hip['minibatches_per_fit'],
                                             'alg_bytes_per_unit': out['roofline']['step_alg_bytes_per_interaction'],
                                             'step_frac': round(hip['interactions_per_s'] * out['roofline']['step_alg_bytes_per_interaction'] / 8e12, 5),
                                             'what': hip['what']} if 'interactions_per_s' in hip else hip)
        if legs is not None:
            # scalars inside `roofline` (record parsers keep its scalar fields) + the object itself LAST on the line (a tail of
            # the line then shows it)
            for name, rec in legs.items():
                if isinstance(rec, dict) and ('frac' in rec or 'step_frac' in rec):
                    out['roofline']['leg_%s_frac' % name] = rec.get('frac', rec.get('step_frac'))
                    out['roofline']['leg_%s_ms_per_step' % name] = rec.get('ms_per_step')
            out['configs'] = legs
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
