"""Backend for the `-m gpu` parity tests: the real csrc/libspotlight_hip.so on cuda:0, torch
tensors as device memory, torch's current stream."""
import numpy as np
import torch

from emu_backend import _Model, _SeqModel
from spotlight_amd import _native


class HipBackend(object):

    def __init__(self):
        assert torch.cuda.is_available()
        self.device = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        self.engine = _native.Engine(0)  # loads libspotlight_hip.so; raises if missing
        self.stream = torch.cuda.current_stream(self.device).cuda_stream

    def alloc(self, a):
        return torch.from_numpy(np.array(a, order='C', copy=True)).to(self.device)

    def ptr(self, t):
        return t.data_ptr() if t is not None else None

    def get(self, t):
        return t.cpu().numpy()

    def model(self, params, opt='adagrad', **hp):
        return _Model(self, params, opt, **hp)

    def seq_model(self, params, opt='adagrad', **hp):
        return _SeqModel(self, params, opt, **hp)

    def close(self):
        torch.cuda.synchronize()
        self.engine.close()
