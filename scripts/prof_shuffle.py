#!/usr/bin/env python
"""Runs slk_shuffle_perm once on n elements (for rocprofv3 --kernel-trace --stats).  usage: prof_shuffle.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
eng = _native.Engine(0)
dev = torch.device('cuda', 0)
perm = torch.empty(n, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
for rep in range(2):
    eng.rng_set_state(np.random.RandomState(rep).get_state())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.shuffle_perm(n, perm.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    print('shuffle_perm(%d): %.1f ms' % (n, (time.perf_counter() - t0) * 1e3))
