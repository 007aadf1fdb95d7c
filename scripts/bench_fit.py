#!/usr/bin/env python
"""End-to-end ImplicitFactorizationModel.fit() throughput ("kernel-only interactions/s AND
end-to-end fit() wall time"), C2-shaped: synthetic uniform ids, 10M users x 1M items, dim 64, bpr,
Adagrad, batch 2^20.  Prints the wall time of fit() per epoch with the epoch shuffle computed on the
device (slk_shuffle_perm, numpy-exact) and the time numpy's own shuffle of the same ids takes on the host.
usage: python scripts/bench_fit.py [n_interactions]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    U, I = 10_000_000, 1_000_000
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32),
                         num_users=U, num_items=I)
    opt = lambda p: torch.optim.Adagrad(p, lr=1e-2)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=1 << 20, optimizer_func=opt,
                                       use_cuda=True, sparse=True, random_state=np.random.RandomState(1))
    t0 = time.perf_counter()
    model.fit(inter)  # includes table initialisation, scratch allocation, the id upload
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    model._n_iter = 3
    t0 = time.perf_counter()
    model.fit(inter)
    torch.cuda.synchronize()
    t_fit = (time.perf_counter() - t0) / 3
    # the host shuffle the reference (and round-1 first half of this package) runs per epoch
    order = np.arange(n)
    t0 = time.perf_counter()
    np.random.RandomState(1).shuffle(order)
    u = inter.user_ids[order]
    t_np = time.perf_counter() - t0
    # the device shuffle alone
    from spotlight_amd.factorization import implicit as host
    dev = model._net.tables()[0].device
    eng = host._engine_for(dev)
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    eng.rng_set_state(np.random.RandomState(1).get_state())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.shuffle_perm(n, perm.data_ptr(), stream=host._stream_for(dev))
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    assert np.array_equal(perm.cpu().numpy(), order)
    print(json.dumps({'interactions': n, 'first_fit_s': t_first, 'fit_s_per_epoch': t_fit,
                      'fit_interactions_per_s': n / t_fit, 'numpy_host_shuffle_s': t_np,
                      'device_shuffle_s': t_dev, 'device_shuffle_fixpoint_sweeps': None,
                      'note': 'fit() per epoch = device shuffle + 2 gathers + training kernels + one D2H of the epoch loss; '
                              'the permutation equals numpy\'s bit for bit (asserted above)'}))


if __name__ == '__main__':
    main()
