#!/usr/bin/env python
"""Random-row regime probe (slk_probe_random_rows): separate [param][state] tables vs interleaved [param|state] rows, at the
per-GPU item-table size of the 1B-item configuration (125M rows x dim 64 = 32 GB + 32 GB) and at C2's (1M rows).
usage: python scripts/probe_layout.py [rows ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd import _native  # noqa: E402

dev = torch.device('cuda', 0)
eng = _native.Engine(0)
D = 64
for rows in [int(x) for x in sys.argv[1:]] or [1_000_000, 12_500_000, 125_000_000]:
    buf = torch.empty(rows * 2 * D, device=dev)
    buf.normal_(0, 0.1)
    n = 1 << 21  # accesses per launch: the item occurrences of one 2^20 minibatch
    rec = {'rows': rows, 'dim': D, 'accesses_per_launch': n, 'bytes_per_access': {'read': 8 * D, 'rmw': 16 * D}}
    for order, oname in ((0, 'ascending'), (1, 'random')):
        for rmw, rname in ((0, 'read'), (1, 'rmw')):
            for layout, lname in ((0, 'separate'), (1, 'interleaved')):
                ms = eng.probe_random_rows(buf.data_ptr(), rows, D, layout, order, rmw, min(n, rows), iters=10,
                                           stream=torch.cuda.current_stream(dev).cuda_stream)
                gb = min(n, rows) * (16 if rmw else 8) * D / ms / 1e6
                rec['%s_%s_%s' % (oname, rname, lname)] = {'ms': ms, 'GBs': gb}
    print(json.dumps(rec), flush=True)
    del buf
    torch.cuda.empty_cache()
