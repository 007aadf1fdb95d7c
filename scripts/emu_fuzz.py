#!/usr/bin/env python
"""Memory-safety fuzz of the engine through the GPU-less harness (test infrastructure): random shapes -- odd embedding dims, one-row
tables, minibatches of 1 / 63 / 65 / 1025 / 3000, every loss and optimizer -- through training, PoolNet, bloom layers, explicit
feedback, the user-row ping-pong scope, the shuffle, the sampler and the fused ranks.  Meant to run under AddressSanitizer:

    SLK_EMU_CXXFLAGS="-fsanitize=address -fno-omit-frame-pointer" LD_PRELOAD="<libasan.so> <libstdc++.so.6>" \
        ASAN_OPTIONS=detect_leaks=0 python scripts/emu_fuzz.py <seed> <seconds>

ASan aborts the process on any out-of-bounds access.  The checks' numeric asserts are IGNORED here: at random shapes they are
ill-posed (hinge gradients cancel to order-dependent residues that Adagrad / Adam normalise to O(lr); a single near-zero score has no
meaningful relative error) -- the tests proper choose their shapes and bounds.  Round 5: 60 000 cases over two rounds, no report."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import engine_checks as ec
from emu_backend import EmuBackend
be = EmuBackend()
import torch  # noqa: E402 -- the embedding front-end (spotlight_amd/embedding.py) asks the host module for its engine
from spotlight_amd.factorization import implicit as _host  # noqa: E402
_host._engine_for = lambda device: be.engine
_host._stream_for = lambda device: 0
_host._model_device = lambda: torch.device('cpu')
rs = np.random.RandomState(int(sys.argv[1]))
budget = float(sys.argv[2])
t0 = time.time(); n = 0; errs = 0
Ds = [1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128, 192, 256]
pick = lambda xs: xs[rs.randint(len(xs))]
while time.time() - t0 < budget:
    kind = rs.randint(11) if len(sys.argv) < 4 else int(sys.argv[3])  # (third argument: one kind only)
    seed = int(rs.randint(1 << 30))
    cfg = None
    try:
        if kind == 0:
            loss, opt, D = pick(ec.ALL_LOSSES), pick(ec.ALL_OPTS), pick(Ds)
            U, I = int(pick([1, 2, 7, 37, 100, 1000])), int(pick([1, 2, 5, 29, 300, 2000]))
            B = int(pick([1, 2, 63, 64, 65, 100, 256, 257, 1000, 1025, 3000])); N = int(max(1, min(pick([1, B, B + 1, 2 * B + 7, 3 * B - 1]), 5000)))
            cfg = ('train', loss, opt, D, U, I, N, B)
            ec.check_train_matches_oracle(be, loss, opt, D, U=U, I=I, N=N, B=B, nn=int(pick([1, 2, 5, 8])), epochs=1, seed=seed, degenerate=True)
        elif kind == 1:
            loss, opt, D = pick(ec.ALL_LOSSES), pick(['adagrad', 'sparse_adam', 'adam_dense']), pick([4, 8, 16, 32, 64, 128])
            I, L, B = int(pick([2, 5, 31, 300])), int(pick([2, 9, 33, 64, 100])), int(pick([1, 3, 16, 40]))
            N = int(pick([2, B + 1, 2 * B + 3]))
            cfg = ('seq', loss, opt, D, I, N, L, B)
            ec.check_seq_train_matches_oracle(be, loss, opt, D, I=I, N=N, L=L, B=B, nn=int(pick([1, 2, 5])), epochs=1, seed=seed)
        elif kind == 2:
            loss, opt, D = pick(ec.ALL_LOSSES), pick(['adagrad', 'sparse_adam']), pick([4, 8, 16, 32, 64, 128])
            ub, ib = int(pick([0, 1, 4, 8])), int(pick([0, 2, 4]))
            cfg = ('bloom', loss, opt, D, ub, ib)
            ec.check_bloom_train_matches_oracle(be, loss, opt, D, user_bloom=ub, item_bloom=ib, U=int(pick([10, 45, 300])), I=int(pick([12, 60, 500])),
                                                N=int(pick([1, 64, 170, 700])), B=int(pick([1, 64, 257])), nn=int(pick([1, 3])), seed=seed)
        elif kind == 3:
            loss, opt, D = pick(ec.EXPLICIT_LOSSES), pick(ec.ALL_OPTS), pick(Ds)
            cfg = ('explicit', loss, opt, D)
            ec.check_explicit_train_matches_oracle(be, loss, opt, D, U=int(pick([1, 37, 500])), I=int(pick([1, 29, 700])), N=int(pick([1, 63, 300, 2000])),
                                                   B=int(pick([1, 64, 257, 1025])), epochs=1, seed=seed)
        elif kind >= 9:
            # training inside a user-row ping-pong scope (two copies of the user table, flag bytes, {g, src} pairs), with and
            # without the item-bias shadow: hot users / items (long runs + both stitch kernels), several chunks, odd dims
            loss, opt, D = pick(['pointwise', 'bpr', 'hinge']), pick(['adagrad', 'sparse_adam', 'sgd']), pick(Ds)
            U, I = int(pick([1, 2, 7, 37, 100, 1000])), int(pick([1, 2, 5, 29, 300, 2000]))
            B = int(pick([1, 2, 63, 64, 65, 100, 256, 257, 1000, 1025, 3000])); N = int(max(1, min(pick([1, B, B + 1, 2 * B + 7, 3 * B - 1]), 6000)))
            opts = pick([None, {'user_lat_max_batch': 0, 'item_lat_max_tiles': 0}, {'item_long_gate': 0}, {'chunk_interactions': max(B, 300)}])
            if rs.randint(2):  # ... and with the single-occurrence fast path forced (Adagrad takes it, the others ignore the option)
                opts = dict(opts or {}, item_single_min_items=1, user_lat_max_batch=int(pick([0, 1 << 17])))
            cfg = ('pingpong', loss, opt, D, U, I, N, B, opts)
            ec.check_user_pingpong_is_bit_neutral(be, loss, opt, D, U=U, I=I, N=N, B=B, seed=seed, options=opts,
                                                  with_bias_shadow=(opt == 'adagrad' and bool(rs.randint(2))), calls=int(pick([1, 2, 3])), sanity=False)
        elif kind == 4:
            n_ids = int(pick([1, 2, 5, 100, 4095, 4096, 4097, 10000, 70000]))
            cfg = ('shuffle', n_ids)
            ec.check_shuffle_matches_numpy(be, n_ids, seed % 1000, rows=int(pick([0, 0, 3])))
        elif kind == 5:
            ni = int(pick([1, 2, 3, 50, 1682, 10 ** 6, 2 ** 31, 2 ** 32]))
            cfg = ('sampler', ni)
            ec.check_sampler_bit_exact(be, ni, counts=(int(pick([1, 5, 623, 624, 625, 3000, 40000])), int(pick([1, 700]))))
        elif kind == 7:
            n_int, nu = int(pick([1, 2, 50, 700, 5000])), int(pick([1, 3, 40, 900]))
            L = int(pick([1, 2, 5, 10, 33]))
            cfg = ('to_sequence', n_int, nu, L)
            ec.check_to_sequence(be, n_int, nu, int(pick([2, 30, 5000])), pick(['int32', 'int64_wide', 'negative', 'float']), L,
                                 pick([None, 1, min(3, L)]), pick([None, 1, 2, L + 3]), seed=seed % 1000)
        elif kind == 8:
            import torch
            from spotlight_amd.embedding import lookup
            rows, dim = int(pick([1, 2, 37, 1000])), int(pick(Ds))
            shape = pick([(1,), (7,), (9, 13), (64, 5), (3000,)])
            cfg = ('lookup', rows, dim, shape)
            w = torch.from_numpy(rs.normal(size=(rows, dim)).astype(np.float32)).requires_grad_(True)
            ids = torch.from_numpy(rs.randint(0, rows, shape))
            out = lookup(w, ids, padding_idx=pick([None, 0]), sparse=bool(rs.randint(2)))
            out.sum().backward()
        else:
            D = pick([4, 8, 24, 64, 128])
            cfg = ('ranks', D)
            ec.check_fused_ranks(be, D=D, U=int(pick([40, 90, 400])), I=int(pick([33, 333, 1500])), n_rows=int(pick([8, 9, 150])), seed=seed)
    except AssertionError as e:
        errs += 1
        if cfg and cfg[0] == 'pingpong':  # bit-neutrality is well-posed at ANY shape: a failed assert here is a bug, not an ill-posed bound
            print('BITDIFF', cfg, seed, repr(e)[:300], flush=True)
    except Exception as e:  # an engine refusal (SlkError) for an unsupported combination is fine; anything else is printed
        print('EXC', cfg, repr(e)[:200], flush=True)
    n += 1
print('ran', n, 'cases;', errs, 'numeric asserts (ignored here)')
