"""TEST HARNESS: drives the engine's C ABI through the fiber-emulator build of the same
sources (tests/emu), with numpy arrays standing in for device memory.  This checks kernel
indexing / sort-key / segment logic and the host control flow on a box without a GPU; the
`-m gpu` tests run the real gfx950 library through the very same checks."""
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


def ptr(a):
    return a.ctypes.data if a is not None else None


class HostModel(object):
    """fp32 numpy tables + optimizer state laid out for the C ABI (emulator: host memory)."""

    def __init__(self, params, opt='adagrad', state1=None, state2=None, **hp):
        self.p = [np.array(x, dtype=np.float32, order='C', copy=True) for x in params]
        self.p[2] = self.p[2].reshape(-1)
        self.p[3] = self.p[3].reshape(-1)
        self.s1 = [np.array(s, np.float32, copy=True) for s in state1] if state1 else [np.zeros_like(x) for x in self.p]
        self.s2 = [np.array(s, np.float32, copy=True) for s in state2] if state2 else [np.zeros_like(x) for x in self.p]
        U, D = self.p[0].shape
        self.tables = _native.make_tables([ptr(x) for x in self.p], U, self.p[1].shape[0], D)
        self.optim = _native.make_optim(opt, [ptr(x) for x in self.s1], [ptr(x) for x in self.s2], **hp)
