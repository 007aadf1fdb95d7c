"""Small host helpers with the names and behaviour of spotlight/torch_utils.py:6-69, for code
written against the reference (the models of this package do their per-epoch shuffle on the device:
`spotlight_amd.factorization.implicit.device_epoch_shuffle`, numpy-exact).

* `gpu` / `cpu`: under PyTorch-ROCm `.cuda()` IS the MI355X.
* `minibatch`: contiguous slices, short last slice; a bare slice (not a 1-tuple) for one input.
* `shuffle`: ONE legacy RandomState.shuffle of arange(n), applied to every array -- the exact
  consumption of the MT19937 stream the rest of the path continues from.
"""
import numpy as np
import torch

_TARGET_GRAD_MESSAGE = ("nn criterions don't compute the gradient w.r.t. targets - please "
                        "mark these variables as volatile or not requiring gradients")


def gpu(tensor, gpu=False):
    if not gpu:
        return tensor
    return tensor.cuda()


def cpu(tensor):
    if tensor.is_cuda:
        return tensor.cpu()
    return tensor


def minibatch(*tensors, **kwargs):
    size = kwargs.get('batch_size', 128)
    single = len(tensors) == 1
    for lo in range(0, len(tensors[0]), size):
        window = slice(lo, lo + size)
        yield tensors[0][window] if single else tuple(x[window] for x in tensors)


def shuffle(*arrays, **kwargs):
    lengths = {len(x) for x in arrays}
    if len(lengths) != 1:
        raise ValueError('All inputs to shuffle must have the same length.')
    rng = kwargs.get('random_state') or np.random.RandomState()
    permutation = np.arange(lengths.pop())
    rng.shuffle(permutation)
    picked = tuple(x[permutation] for x in arrays)
    return picked[0] if len(picked) == 1 else picked


def assert_no_grad(variable):
    if variable.requires_grad:
        raise ValueError(_TARGET_GRAD_MESSAGE)


def set_seed(seed, cuda=False):
    torch.manual_seed(seed)
    if cuda and torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
