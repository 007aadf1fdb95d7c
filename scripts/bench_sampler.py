"""Times slk_sample_items alone (GPU): count draws in [0, num_items)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from spotlight_amd import _native
eng = _native.Engine(0)
eng.rng_set_state(np.random.RandomState(3).get_state())
for count in (1 << 20, 1 << 23):
    out = torch.empty(count, dtype=torch.int64, device='cuda')
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sample_items(10 ** 6, count, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('count %d rep %d: %.3f ms  (%.2f G draws/s)' % (count, rep, dt * 1e3, count / dt / 1e9))
eng.profile_enable(True)
