#!/usr/bin/env python
"""The drop-in fit() at the C2 tables, 2^25 interactions x n_iter epochs, for a rocprofv3 --kernel-trace run
(scripts/fit_epoch_breakdown.py reads the trace).   usage: python scripts/trace_fit_epochs.py [n] [epochs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.interactions import Interactions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    U, I = 10_000_000, 1_000_000
    rs = np.random.RandomState(0)
    inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
    model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=1 << 20, use_cuda=True, sparse=True,
                                       optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2), random_state=np.random.RandomState(1))
    model.fit(inter)
    torch.cuda.synchronize()
    model._n_iter = epochs
    t0 = time.perf_counter()
    model.fit(inter)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('fit: %d interactions x %d epochs: %.2f ms per epoch, %.3f G interactions/s' % (n, epochs, dt / epochs * 1e3, n * epochs / dt / 1e9))


if __name__ == '__main__':
    main()
