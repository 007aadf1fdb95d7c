"""Host glue mirroring spotlight/torch_utils.py:6-69.

`gpu()` moves to the HIP device (`tensor.cuda()` IS the MI355X under PyTorch-ROCm);
`shuffle()` stays on the host and consumes the numpy RandomState stream exactly like the
reference, because the negatives drawn afterwards continue the same MT19937 stream.
"""
import numpy as np
import torch


def gpu(tensor, gpu=False):
    return tensor.cuda() if gpu else tensor


def cpu(tensor):
    return tensor.cpu() if tensor.is_cuda else tensor


def minibatch(*tensors, **kwargs):
    batch_size = kwargs.get('batch_size', 128)
    n = len(tensors[0])
    for start in range(0, n, batch_size):
        if len(tensors) == 1:
            yield tensors[0][start:start + batch_size]
        else:
            yield tuple(t[start:start + batch_size] for t in tensors)


def shuffle(*arrays, **kwargs):
    random_state = kwargs.get('random_state')
    if len({len(a) for a in arrays}) != 1:
        raise ValueError('All inputs to shuffle must have the same length.')
    if random_state is None:
        random_state = np.random.RandomState()
    order = np.arange(len(arrays[0]))
    random_state.shuffle(order)
    if len(arrays) == 1:
        return arrays[0][order]
    return tuple(a[order] for a in arrays)


def assert_no_grad(variable):
    if variable.requires_grad:
        raise ValueError("nn criterions don't compute the gradient w.r.t. targets - please "
                         "mark these variables as volatile or not requiring gradients")


def set_seed(seed, cuda=False):
    torch.manual_seed(seed)
    if cuda and torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
