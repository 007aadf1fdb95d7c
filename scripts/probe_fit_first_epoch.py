#!/usr/bin/env python
"""What does the FIRST epoch of a large fit() cost beyond a steady-state one?  Host-side pieces timed alone (id checks,
host -> HBM upload in 1 .. 8 slices on as many threads), then fit() with 1, 2, 3, 10 epochs."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel, IdUpload
from spotlight_amd.interactions import Interactions

n, U, I, B = 1 << 25, 10_000_000, 1_000_000, 1 << 20
dev = torch.device('cuda', 0)
rs = np.random.RandomState(5)
inter = Interactions(rs.randint(0, U, n).astype(np.int32), rs.randint(0, I, n).astype(np.int32), num_users=U, num_items=I)
model = ImplicitFactorizationModel(loss='bpr', embedding_dim=64, n_iter=1, batch_size=B, use_cuda=True, sparse=True,
                                   optimizer_func=lambda p: torch.optim.Adagrad(p, lr=1e-2), random_state=np.random.RandomState(1))
model.fit(inter); torch.cuda.synchronize()


def t(f, reps=3):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) * 1e3, 2))
    return out


print('check_input ms', t(lambda: model._check_input(inter.user_ids, inter.item_ids)))
print('IdUpload ms', t(lambda: IdUpload([inter.user_ids, inter.item_ids], dev).result()))
for k in (1, 2, 4, 8):
    dst = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]

    def up():
        th = []
        for a, d in zip((inter.user_ids, inter.item_ids), dst):
            for j in range(k):
                lo, hi = j * n // k, (j + 1) * n // k
                def w(a=a, d=d, lo=lo, hi=hi):
                    torch.cuda.set_device(dev)
                    d[lo:hi].copy_(torch.from_numpy(a[lo:hi]))
                th.append(threading.Thread(target=w)); th[-1].start()
        for x in th:
            x.join()
    print('upload in %d slices per array ms' % k, t(up))
pin = [torch.from_numpy(a).pin_memory() for a in (inter.user_ids, inter.item_ids)]
print('pinned upload ms', t(lambda: [d.copy_(p, non_blocking=True) for d, p in zip(dst, pin)]))
def traced(e):
    model._n_iter = e
    model._fit_timeline = []
    r0 = torch.cuda.memory_reserved()
    t0 = time.perf_counter(); model.fit(inter); torch.cuda.synchronize(); t1 = time.perf_counter()
    keep = [(l, x) for l, x in model._fit_timeline if not l.startswith(('train', 'shuffle 2', 'shuffle 3', 'prefetch 2', 'prefetch 3')) or e <= 2]
    print('n_iter=%d total %.2f ms (reserved %.2f -> %.2f GB): ' % (e, (t1 - t0) * 1e3, r0 / 1e9, torch.cuda.memory_reserved() / 1e9)
          + ', '.join('%s +%.1f' % (l, (x - t0) * 1e3) for l, x in keep[:14]))


for e in (1, 2, 4, 1, 2, 1, 10, 10, 1, 10):
    traced(e)
