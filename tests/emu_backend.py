"""TEST HARNESS: drives the engine's C ABI through the fiber-emulator build of the same
sources (tests/emu), with numpy arrays standing in for device memory.  This checks kernel
indexing / sort-key / segment logic and the host control flow on a box without a GPU; the
`-m gpu` tests run the real gfx950 library through the very same checks
(tests/hip_backend.py)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))

from spotlight_amd import _native  # noqa: E402

_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        import build_emu
        _EMU = _native.bind(C.CDLL(build_emu.build()))
    return _EMU


def _bloom_struct(desc, rows):
    """oracle.bloom_desc() dict -> SlkBloom for a compressed table with `rows` rows."""
    if desc is None:
        return None
    return _native.make_bloom(rows, desc['n_hash'], padding_idx=desc['padding_idx'],
                              skip_row=None if desc['skip_row'] < 0 else desc['skip_row'], seeds=desc['seeds'])


class _Model(object):
    def __init__(self, be, params, opt='adagrad', user_bloom=None, item_bloom=None, **hp):
        f = lambda x: be.alloc(np.array(x, dtype=np.float32, order='C'))
        self.p = [f(params[0]), f(params[1]), f(np.asarray(params[2]).reshape(-1)), f(np.asarray(params[3]).reshape(-1))]
        self.s1 = [be.alloc(np.zeros(be.get(x).shape, np.float32)) for x in self.p]
        self.s2 = [be.alloc(np.zeros(be.get(x).shape, np.float32)) for x in self.p]
        D = be.get(self.p[0]).shape[1]
        U, I = be.get(self.p[2]).shape[0], be.get(self.p[3]).shape[0]  # id ranges = bias rows
        self.tables = _native.make_tables([be.ptr(x) for x in self.p], U, I, D,
                                          user_bloom=_bloom_struct(user_bloom, be.get(self.p[0]).shape[0]),
                                          item_bloom=_bloom_struct(item_bloom, be.get(self.p[1]).shape[0]))
        self.optim = _native.make_optim(opt, [be.ptr(x) for x in self.s1], [be.ptr(x) for x in self.s2], **hp)


class _SeqModel(object):
    """PoolNet tables (item_embeddings, item_biases) + optimizer state in ABI slots 1 and 3."""

    def __init__(self, be, params, opt='adagrad', item_bloom=None, **hp):
        f = lambda x: be.alloc(np.array(x, dtype=np.float32, order='C'))
        self.p = [f(params[0]), f(np.asarray(params[1]).reshape(-1))]
        self.s1 = [be.alloc(np.zeros(be.get(x).shape, np.float32)) for x in self.p]
        self.s2 = [be.alloc(np.zeros(be.get(x).shape, np.float32)) for x in self.p]
        rows, D = be.get(self.p[0]).shape
        I = be.get(self.p[1]).shape[0]  # id range = bias rows (the embedding table may be a compressed one)
        self.tables = _native.make_seq_tables(be.ptr(self.p[0]), be.ptr(self.p[1]), I, D,
                                              item_bloom=_bloom_struct(item_bloom, rows))
        slot = lambda xs: [None, be.ptr(xs[0]), None, be.ptr(xs[1])]
        self.optim = _native.make_optim(opt, slot(self.s1), slot(self.s2), **hp)


class EmuBackend(object):
    stream = 0

    def __init__(self):
        self.engine = _native.Engine(0, lib=emu_lib())

    def alloc(self, a):
        return np.array(a, order='C', copy=True)

    def ptr(self, a):
        return a.ctypes.data if a is not None else None

    def get(self, a):
        return a

    def model(self, params, opt='adagrad', **hp):
        return _Model(self, params, opt, **hp)  # hp may carry user_bloom / item_bloom descriptors

    def seq_model(self, params, opt='adagrad', **hp):
        return _SeqModel(self, params, opt, **hp)

    def close(self):
        self.engine.close()
