"""Diagnostic (GPU): where do engine and oracle parameters differ after training?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from hip_backend import HipBackend
from oracle.oracle import BilinearOracle, Rng

be = HipBackend()
eng = be.engine


def run(loss, opt, D, U, I, N, B, nn=3, epochs=1, seed=5, scale=0.3):
    rs = np.random.RandomState(seed)
    users = rs.randint(0, U, N).astype(np.int64)
    items = rs.randint(0, I, N).astype(np.int64)
    params = [rs.normal(0, scale, (U, D)), rs.normal(0, scale, (I, D)), rs.normal(0, 0.1, U), rs.normal(0, 0.1, I)]
    hp = dict(lr=0.05)
    ora = BilinearOracle(*params, opt=opt, sparse_grads=True, **hp)
    dev = be.model(params, opt=opt, **hp)
    state = np.random.RandomState(9).get_state()
    orng = Rng(state=state)
    eng.rng_set_state(state)
    n_mb = (N + B - 1) // B
    d_users, d_items = be.alloc(users), be.alloc(items)
    for e in range(epochs):
        wl = ora.train(orng, users, items, B, loss=loss, n_neg=nn)
        mb = be.alloc(np.zeros(n_mb, dtype=np.float32))
        eng.bilinear_train(dev.tables, dev.optim, be.ptr(d_users), be.ptr(d_items), N, B, loss, nn, be.ptr(mb),
                           stream=be.stream)
        gl = be.get(mb)
        print('  loss rel err per mb', np.abs(gl - wl) / np.abs(wl))
    for t in range(4):
        a, b = be.get(dev.p[t]).astype(np.float64).ravel(), ora.p[t].astype(np.float64).ravel()
        init = np.asarray(params[t], np.float32).astype(np.float64).ravel()
        d = np.abs(a - b)
        sc = np.abs(b).max()
        w = d.argmax()
        print('  table %d: max|d|/max|p| %.2e  frac>1e-5 %.2e frac>1e-4 %.2e frac>1e-3 %.2e | worst: got %.6g want %.6g init %.6g'
              % (t, d.max() / sc, (d > 1e-5 * sc).mean(), (d > 1e-4 * sc).mean(), (d > 1e-3 * sc).mean(),
                 a[w], b[w], init[w]))
        sa, sb = be.get(dev.s1[t]).astype(np.float64).ravel(), ora.s1[t].astype(np.float64).ravel()
        ds = np.abs(sa - sb)
        print('           state1: max rel %.2e ; at worst param elem: got %.6g want %.6g' % (ds.max() / max(np.abs(sb).max(), 1e-30), sa[w], sb[w]))


for cfg in [('bpr', 'adagrad', 256, 23, 31, 27, 20), ('bpr', 'adagrad', 64, 3000, 1000, 50000, 8192),
            ('bpr', 'sparse_adam', 64, 3000, 1000, 50000, 8192), ('bpr', 'adagrad', 64, 3000, 1000, 8192, 8192)]:
    print(cfg)
    run(*cfg)
