"""Negative sampling (mirrors spotlight/sampling.py:8-36).

`sample_items` is the host numpy form kept for API compatibility.  During fit() the draw
happens on the GPU (csrc/slk_rng.hip) from the very same MT19937 stream, bit-exactly;
`sample_items_device` exposes that path directly.
"""
import numpy as np


def sample_items(num_items, shape, random_state=None):
    if random_state is None:
        random_state = np.random.RandomState()
    return random_state.randint(0, num_items, shape, dtype=np.int64)


def sample_items_device(num_items, shape, random_state, engine=None, device=None):
    """Same ids as `sample_items`, drawn by the gfx950 sampler; returns an int64 tensor on the
    HIP device and advances `random_state` exactly as numpy would have."""
    import torch

    from spotlight_amd import _native
    engine = engine or _native.Engine(torch.cuda.current_device())
    count = int(np.prod(shape))
    out = torch.empty(count, dtype=torch.int64, device=device or 'cuda')
    engine.rng_set_state(random_state.get_state())
    engine.sample_items(num_items, count, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    random_state.set_state(engine.rng_get_state())
    return out.view(shape)
