#!/usr/bin/env python
"""fit() end to end at C2 shapes under the host-side choices that are cheap to flip (profiles/r03_*):
  * epochs prepared ahead on a second slk_ctx / stream (ImplicitFactorizationModel._fit_pipelined) or in line,
  * how long the id upload takes by itself.
usage: python scripts/bench_fit_modes.py [n_interactions] [n_iter]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization import implicit as host  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
    n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    U, I = 10_000_000, 1_000_000
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
    dev = torch.device('cuda', 0)
    out = {'interactions': n, 'n_iter': n_iter}
    # the upload by itself (pageable int32 -> int64 on the device), twice
    for k in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = host.ids_to_device(inter.user_ids, dev)
        b = host.ids_to_device(inter.item_ids, dev)
        torch.cuda.synchronize()
        out['upload_s_%d' % k] = time.perf_counter() - t0
        del a, b
    opt = lambda p: torch.optim.Adagrad(p, lr=1e-2)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=1 << 20, optimizer_func=opt,
                                       use_cuda=True, sparse=True, random_state=np.random.RandomState(1))
    model.fit(inter)  # table initialisation, scratch
    torch.cuda.synchronize()
    model._n_iter = n_iter
    # where the host spends the wall time of fit(): every engine call (both lanes) timed on the host
    eng = host._engine_for(dev)
    prep, _ = host._prep_lane_for(dev)
    acc = {}

    def wrap(obj, name, tag):
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[tag] = acc.get(tag, 0.0) + time.perf_counter() - t0
        setattr(obj, name, g)
    for nm in ('rng_set_state', 'rng_get_state', 'rng_get_state_sampled', 'shuffle_perm', 'gather_rows_i64', 'bilinear_train',
               'bilinear_reserve'):
        wrap(eng, nm, 'train_lane.' + nm)
        if prep is not eng:
            wrap(prep, nm, 'prep_lane.' + nm)
    for label, max_draws in (('in_line', 1 << 22), ('prepared_ahead', 1 << 40), ('in_line_again', 1 << 22)):
        acc.clear()
        host._PIPELINE_MAX_DRAWS = max_draws
        t0 = time.perf_counter()
        model.fit(inter)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_iter
        out[label] = {'s_per_epoch': dt, 'G_interactions_per_s': n / dt / 1e9,
                      'host_ms_per_epoch_by_call': {k: round(v / n_iter * 1e3, 3) for k, v in sorted(acc.items())}}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
