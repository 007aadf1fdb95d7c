#!/usr/bin/env python
"""Micro-benchmark of the engine's radix sort (slk_probe_sort) at the training prep's shapes: one JSON line per case with
the average ms, the pairs/s and the bytes/s on the sort's own traffic (per pass: pair read + written; + 4 B/pair histogram).

    python scripts/bench_sort.py [--out gpurun_out/sort.jsonl]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spotlight_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--cases', default='', help='comma-separated case indices (default: all)')
    ap.add_argument('--cfgs', default='1,0', help='tile shapes to time: 1 = 512 threads x 16 keys (what sorts of >= 2^20 pairs use), 0 = 256 x 16')
    ap.add_argument('--set', action='append', default=[], metavar='NAME=VALUE', help='engine option, e.g. sort_big_min=1')
    ap.add_argument('--debug-sweep', action='store_true', help='also time the measurement-only forms (sort_debug 1, 2, 3: wrong results)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    eng = _native.Engine(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    for kv in args.set:
        eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    cases = [  # (label, kind, n, bits, seg_len)
        ('user side C2: 8 x 2^20 pairs u32+u64, 24 bits, segmented', 1, 8 << 20, 24, 1 << 20),
        ('user side C2 as round 3 sorted it: 27 bits, one array', 1, 8 << 20, 27, 0),
        ('item side C2: 8 x 2^21 pairs u32+u32, 20 bits, segmented', 0, 16 << 20, 20, 2 << 20),
        ('item side C2 as round 3 sorted it: 23 bits, one array', 0, 16 << 20, 23, 0),
        ('item side C5 shard: 8 x 2^21 pairs, 27 bits, segmented', 0, 16 << 20, 27, 2 << 20),
        ('C4 PoolNet: 4 x 1.6 M pairs, 20 bits, segmented', 0, 4 * 1638400, 20, 1638400),
        ('C3 live list: 2^19 pairs, 21 bits', 0, 1 << 19, 21, 0),
        ('one minibatch of 65 536 x 2 pairs, 20 bits', 0, 1 << 17, 20, 0),
        ('4096 pairs (single tile), 20 bits', 0, 4096, 20, 0),
    ]
    out = open(args.out, 'a') if args.out else None
    if args.cases:
        cases = [cases[int(i)] for i in args.cases.split(',')]
    for cfg, dbg in [(int(c), 0) for c in args.cfgs.split(',')] + ([(1, 1), (1, 2), (1, 3)] if args.debug_sweep else []):
        eng.set_option('sort_big_min', 1 if cfg else 1 << 62)
        eng.set_option('sort_debug', dbg)
        for label, kind, n, bits, seg in cases:
            if (cfg != 1 or dbg) and n < (1 << 20):
                continue
            kt = torch.int64 if kind == 2 else torch.int32
            vt = torch.int64 if kind == 1 else torch.int32
            keys = torch.randint(0, 1 << min(bits, 30), (n,), device=dev, dtype=torch.int64).to(kt)
            vals = torch.arange(n, device=dev, dtype=torch.int64).to(vt)
            ko, vo = torch.empty_like(keys), torch.empty_like(vals)
            ms = eng.probe_sort(kind, keys.data_ptr(), ko.data_ptr(), vals.data_ptr(), vo.data_ptr(), n, bits, seg_len=seg,
                                iters=args.iters, stream=st)
            passes = (bits + 7) // 8
            pair = keys.element_size() + vals.element_size()
            traffic = n * (passes * 2 * pair + keys.element_size())
            rec = {'case': label, 'options': args.set, 'big_tiles': cfg, 'sort_debug': dbg, 'n': n, 'bits': bits, 'seg_len': seg, 'passes': passes, 'ms': round(ms, 4),
                   'gpairs_per_s': round(n / ms / 1e6, 3), 'tb_per_s_own_traffic': round(traffic / ms / 1e9, 3)}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + '\n')
    eng.set_option('sort_debug', 0)
    eng.set_option('sort_big_min', 1 << 20)
    eng.close()


if __name__ == '__main__':
    main()
