"""ctypes front-end of oracle/slk_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It restates, on the CPU and in scalar C, the reference path
spotlight/factorization/implicit.py:184-275 (+ sampling.py, torch_utils.py:35-52,
layers.py:178-204, losses.py:18-166 and the torch optimisers the reference
instantiates).  Pinned against the live reference by oracle/make_golden.py and against
tests/golden/*.npz by tests/test_oracle.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libslk_oracle.so')

EXPLICIT_LOSSES = {'regression': 4, 'poisson': 5, 'logistic': 6}
LOSSES = {'pointwise': 0, 'bpr': 1, 'hinge': 2, 'adaptive_hinge': 3}
OPTS = {'adagrad': 0, 'sparse_adam': 1, 'adam_dense': 2, 'adagrad_dense': 3, 'sgd': 4}


def build(force=False):
    src = os.path.join(_HERE, 'slk_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B', '_build/libslk_oracle.so'])
    return _SO


class _Model(C.Structure):
    _fields_ = [('p', C.c_void_p * 4), ('s1', C.c_void_p * 4), ('s2', C.c_void_p * 4),
                ('num_users', C.c_int64), ('num_items', C.c_int64),
                ('dim', C.c_int32), ('opt_kind', C.c_int32),
                ('sparse_grads', C.c_int32), ('pad_', C.c_int32),
                ('step', C.c_int64),
                ('lr', C.c_double), ('eps', C.c_double), ('beta1', C.c_double),
                ('beta2', C.c_double), ('weight_decay', C.c_double), ('lr_decay', C.c_double)]


class _Rng(C.Structure):
    _fields_ = [('key', C.c_uint32 * 624), ('pos', C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.slko_sizeof_model() == C.sizeof(_Model)
        assert _lib.slko_sizeof_rng() == C.sizeof(_Rng)
        _lib.slko_rng_next32.restype = C.c_uint32
        _lib.slko_murmur3_32.restype = C.c_int32
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Rng(object):
    """numpy legacy RandomState stream (MT19937 + masked rejection)."""

    def __init__(self, seed=None, state=None):
        self._r = _Rng()
        if state is not None:
            self.set_state(state)
        else:
            lib().slko_rng_seed(C.byref(self._r), C.c_uint32(seed))

    def set_state(self, state):
        """Accepts numpy's RandomState.get_state() tuple."""
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        lib().slko_rng_set(C.byref(self._r), _ptr(key), C.c_int32(int(state[2])))

    def get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        lib().slko_rng_get(C.byref(self._r), _ptr(key), C.byref(pos))
        return ('MT19937', key, int(pos.value), 0, 0.0)

    def next32(self):
        return int(lib().slko_rng_next32(C.byref(self._r)))

    def randint(self, num_items, count):
        out = np.empty(int(count), dtype=np.int64)
        lib().slko_randint(C.byref(self._r), C.c_int64(num_items), C.c_int64(count), _ptr(out))
        return out

    def shuffle_perm(self, n):
        out = np.empty(int(n), dtype=np.int64)
        lib().slko_shuffle_perm(C.byref(self._r), C.c_int64(n), _ptr(out))
        return out


def murmur3_32(key, seed):
    return int(lib().slko_murmur3_32(C.c_int32(int(key)), C.c_uint32(int(seed))))


def bloom_indices(ids, seeds, compressed_rows, padding_idx=0):
    ids = np.ascontiguousarray(ids, dtype=np.int64).ravel()
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    out = np.empty((ids.size, seeds.size), dtype=np.int64)
    lib().slko_bloom_indices(_ptr(ids), C.c_int64(ids.size), _ptr(seeds), C.c_int(seeds.size),
                             C.c_int64(compressed_rows), C.c_int64(padding_idx), _ptr(out))
    return out


class BilinearOracle(object):
    """Holds fp32 numpy copies of the four BilinearNet tables + optimizer state."""

    def __init__(self, user_emb, item_emb, user_bias, item_bias, opt='adagrad', lr=1e-2,
                 eps=None, betas=(0.9, 0.999), weight_decay=0.0, lr_decay=0.0, step=0,
                 state1=None, state2=None, sparse_grads=False):
        f = lambda a: np.array(a, dtype=np.float32, order='C', copy=True)
        self.p = [f(user_emb), f(item_emb), f(user_bias).reshape(-1), f(item_bias).reshape(-1)]
        self.opt = opt
        if eps is None:
            eps = 1e-10 if opt.startswith('adagrad') else 1e-8
        self.s1 = [f(s) for s in state1] if state1 is not None else [np.zeros_like(p) for p in self.p]
        self.s2 = [f(s) for s in state2] if state2 is not None else [np.zeros_like(p) for p in self.p]
        self.m = _Model()
        for t in range(4):
            self.m.p[t] = self.p[t].ctypes.data
            self.m.s1[t] = self.s1[t].ctypes.data
            self.m.s2[t] = self.s2[t].ctypes.data
        self.m.num_users, self.m.dim = self.p[0].shape
        self.m.num_items = self.p[1].shape[0]
        self.m.opt_kind = OPTS[opt]
        self.m.sparse_grads = int(bool(sparse_grads))
        self.m.lr, self.m.eps = lr, eps
        self.m.beta1, self.m.beta2 = betas
        self.m.weight_decay, self.m.lr_decay = weight_decay, lr_decay
        self.m.step = step

    @property
    def step_count(self):
        return int(self.m.step)

    def predict(self, users, items=None):
        users = np.ascontiguousarray(np.atleast_1d(users), dtype=np.int64)
        if items is None:
            n = int(self.m.num_items)
        else:
            items = np.ascontiguousarray(items, dtype=np.int64)
            n = items.size
        out = np.empty(n, dtype=np.float32)
        lib().slko_bilinear_predict(C.byref(self.m), _ptr(users), C.c_int64(users.size),
                                    _ptr(items), C.c_int64(n), _ptr(out))
        return out

    def step(self, users, pos, neg, loss='bpr', n_neg=1, want_grads=False):
        users = np.ascontiguousarray(users, dtype=np.int64)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        neg = np.ascontiguousarray(neg, dtype=np.int64).ravel()
        loss_out = C.c_float()
        dg = None
        dgp = None
        if want_grads:
            dg = [np.zeros_like(p) for p in self.p]
            dgp = (C.c_void_p * 4)(*[g.ctypes.data for g in dg])
        rc = lib().slko_bilinear_step(C.byref(self.m), _ptr(users), _ptr(pos), _ptr(neg),
                                      C.c_int64(users.size), C.c_int(n_neg), C.c_int(LOSSES[loss]),
                                      C.byref(loss_out), dgp)
        assert rc == 0
        return (float(loss_out.value), dg) if want_grads else float(loss_out.value)

    def train(self, rng, users, items, batch_size, loss='bpr', n_neg=1, neg_in=None,
              want_negs=False):
        users = np.ascontiguousarray(users, dtype=np.int64)
        items = np.ascontiguousarray(items, dtype=np.int64)
        n = users.size
        nn = n_neg if loss == 'adaptive_hinge' else 1
        n_mb = (n + batch_size - 1) // batch_size
        mb_loss = np.empty(n_mb, dtype=np.float32)
        neg_out = np.empty(n * nn, dtype=np.int64) if want_negs else None
        if neg_in is not None:
            neg_in = np.ascontiguousarray(neg_in, dtype=np.int64).ravel()
        rc = lib().slko_bilinear_train(C.byref(self.m), C.byref(rng._r) if rng is not None else None,
                                       _ptr(users), _ptr(items), C.c_int64(n),
                                       C.c_int64(batch_size), C.c_int(LOSSES[loss]), C.c_int(n_neg),
                                       _ptr(neg_in), _ptr(neg_out), _ptr(mb_loss))
        assert rc == 0
        return (mb_loss, neg_out) if want_negs else mb_loss

    # -- explicit feedback (spotlight/factorization/explicit.py:173-243, losses.py:169-244) --------
    def explicit_step(self, users, items, ratings, loss='regression', want_grads=False):
        users = np.ascontiguousarray(users, dtype=np.int64)
        items = np.ascontiguousarray(items, dtype=np.int64)
        ratings = np.ascontiguousarray(ratings, dtype=np.float32)
        loss_out = C.c_float()
        dg, dgp = None, None
        if want_grads:
            dg = [np.zeros_like(p) for p in self.p]
            dgp = (C.c_void_p * 4)(*[g.ctypes.data for g in dg])
        rc = lib().slko_explicit_step(C.byref(self.m), _ptr(users), _ptr(items), _ptr(ratings), C.c_int64(users.size),
                                      C.c_int(EXPLICIT_LOSSES[loss]), C.byref(loss_out), dgp)
        assert rc == 0
        return (float(loss_out.value), dg) if want_grads else float(loss_out.value)

    def explicit_train(self, users, items, ratings, batch_size, loss='regression'):
        users = np.ascontiguousarray(users, dtype=np.int64)
        items = np.ascontiguousarray(items, dtype=np.int64)
        ratings = np.ascontiguousarray(ratings, dtype=np.float32)
        n = users.size
        mb_loss = np.empty((n + batch_size - 1) // batch_size, dtype=np.float32)
        rc = lib().slko_explicit_train(C.byref(self.m), _ptr(users), _ptr(items), _ptr(ratings), C.c_int64(n),
                                       C.c_int64(batch_size), C.c_int(EXPLICIT_LOSSES[loss]), _ptr(mb_loss))
        assert rc == 0
        return mb_loss

    def explicit_predict(self, users, items=None, loss='regression'):
        """explicit.py:245-284: exp() of the score for poisson, sigmoid() for logistic."""
        out = self.predict(users, items)
        if loss == 'poisson':
            return np.exp(out)
        if loss == 'logistic':
            return (1.0 / (1.0 + np.exp(-out.astype(np.float32)))).astype(np.float32)
        return out


class _Bloom(C.Structure):
    _fields_ = [('rows', C.c_int64), ('n_hash', C.c_int32), ('pad_', C.c_int32), ('padding_idx', C.c_int64),
                ('skip_row', C.c_int64), ('seeds', C.c_uint32 * 8)]


class _SeqModel(C.Structure):
    _fields_ = [('p', C.c_void_p * 2), ('s1', C.c_void_p * 2), ('s2', C.c_void_p * 2),
                ('num_items', C.c_int64), ('dim', C.c_int32), ('opt_kind', C.c_int32),
                ('padding_idx', C.c_int64), ('step', C.c_int64),
                ('lr', C.c_double), ('eps', C.c_double), ('beta1', C.c_double),
                ('beta2', C.c_double), ('weight_decay', C.c_double), ('lr_decay', C.c_double),
                ('bloom', _Bloom)]


class PoolNetOracle(object):
    """fp32 numpy copies of PoolNet's two tables (item_embeddings, item_biases) + optimizer
    state; restates spotlight/sequence/implicit.py:193-340 with sequence/representations.py:76-144."""

    def __init__(self, item_emb, item_bias, opt='adagrad', lr=1e-2, eps=None, betas=(0.9, 0.999),
                 weight_decay=0.0, lr_decay=0.0, step=0, state1=None, state2=None, padding_idx=0, item_bloom=None):
        """`item_bloom`: bloom_desc(...) when item_embeddings is a BloomEmbedding(num_items, D, padding_idx=0)
        (item_emb is then the compressed table; num_items = len(item_bias))."""
        assert lib().slko_sizeof_seq_model() == C.sizeof(_SeqModel)
        f = lambda a: np.array(a, dtype=np.float32, order='C', copy=True)
        self.p = [f(item_emb), f(item_bias).reshape(-1)]
        if eps is None:
            eps = 1e-10 if opt.startswith('adagrad') else 1e-8
        self.s1 = [f(s) for s in state1] if state1 is not None else [np.zeros_like(p) for p in self.p]
        self.s2 = [f(s) for s in state2] if state2 is not None else [np.zeros_like(p) for p in self.p]
        self.m = _SeqModel()
        for t in range(2):
            self.m.p[t] = self.p[t].ctypes.data
            self.m.s1[t] = self.s1[t].ctypes.data
            self.m.s2[t] = self.s2[t].ctypes.data
        self.m.num_items, self.m.dim = self.p[1].shape[0], self.p[0].shape[1]
        if item_bloom is not None:
            self.m.bloom = _fill_bloom(item_bloom, self.p[0].shape[0])
        self.m.opt_kind = OPTS[opt]
        self.m.padding_idx = -1 if padding_idx is None else int(padding_idx)
        self.m.lr, self.m.eps = lr, eps
        self.m.beta1, self.m.beta2 = betas
        self.m.weight_decay, self.m.lr_decay = weight_decay, lr_decay
        self.m.step = step

    @property
    def step_count(self):
        return int(self.m.step)

    def step(self, seqs, neg, loss='bpr', n_neg=1, want_grads=False):
        seqs = np.ascontiguousarray(seqs, dtype=np.int64)
        B, L = seqs.shape
        neg = np.ascontiguousarray(neg, dtype=np.int64).ravel()
        assert neg.size == B * L * (n_neg if loss == 'adaptive_hinge' else 1)
        loss_out = C.c_float()
        dg, dgp = None, None
        if want_grads:
            dg = [np.zeros_like(p) for p in self.p]
            dgp = (C.c_void_p * 2)(*[g.ctypes.data for g in dg])
        rc = lib().slko_poolnet_step(C.byref(self.m), _ptr(seqs), _ptr(neg), C.c_int64(B), C.c_int64(L),
                                     C.c_int(n_neg), C.c_int(LOSSES[loss]), C.byref(loss_out), dgp)
        assert rc == 0
        return (float(loss_out.value), dg) if want_grads else float(loss_out.value)

    def train(self, rng, seqs, batch_size, loss='bpr', n_neg=1, neg_in=None, want_negs=False):
        seqs = np.ascontiguousarray(seqs, dtype=np.int64)
        n, L = seqs.shape
        nn = n_neg if loss == 'adaptive_hinge' else 1
        n_mb = (n + batch_size - 1) // batch_size
        mb_loss = np.empty(n_mb, dtype=np.float32)
        neg_out = np.empty(n * nn * L, dtype=np.int64) if want_negs else None
        if neg_in is not None:
            neg_in = np.ascontiguousarray(neg_in, dtype=np.int64).ravel()
        rc = lib().slko_poolnet_train(C.byref(self.m), C.byref(rng._r) if rng is not None else None,
                                      _ptr(seqs), C.c_int64(n), C.c_int64(L), C.c_int64(batch_size),
                                      C.c_int(LOSSES[loss]), C.c_int(n_neg), _ptr(neg_in), _ptr(neg_out),
                                      _ptr(mb_loss))
        assert rc == 0
        return (mb_loss, neg_out) if want_negs else mb_loss

    def predict(self, seq, items=None):
        seq = np.ascontiguousarray(np.asarray(seq).reshape(-1), dtype=np.int64)
        if items is None:
            n = int(self.m.num_items)
        else:
            items = np.ascontiguousarray(np.asarray(items).reshape(-1), dtype=np.int64)
            n = items.size
        out = np.empty(n, dtype=np.float32)
        lib().slko_poolnet_predict(C.byref(self.m), _ptr(seq), C.c_int64(seq.size), _ptr(items), C.c_int64(n),
                                   _ptr(out))
        return out


BLOOM_SEEDS = [179424941, 179425457, 179425907, 179426369,
               179424977, 179425517, 179425943, 179426407]  # spotlight/layers.py:13-20, first eight


class _BModel(C.Structure):
    _fields_ = [('p', C.c_void_p * 4), ('s1', C.c_void_p * 4), ('s2', C.c_void_p * 4),
                ('num_users', C.c_int64), ('num_items', C.c_int64), ('dim', C.c_int32), ('opt_kind', C.c_int32),
                ('step', C.c_int64),
                ('lr', C.c_double), ('eps', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('weight_decay', C.c_double), ('lr_decay', C.c_double), ('bloom', _Bloom * 2)]


def bloom_desc(n_hash=4, padding_idx=0, bag=False, seeds=None):
    """Descriptor of a BloomEmbedding (layers.py:134-176); the compressed row count comes from
    the table itself."""
    return dict(n_hash=n_hash, padding_idx=padding_idx, skip_row=-1 if bag else padding_idx,
                seeds=list(seeds if seeds is not None else BLOOM_SEEDS[:n_hash]))


def _fill_bloom(desc, rows):
    b = _Bloom()
    b.rows, b.n_hash = int(rows), desc['n_hash']
    b.padding_idx, b.skip_row = desc['padding_idx'], desc['skip_row']
    for h, s in enumerate(desc['seeds']):
        b.seeds[h] = s
    return b


class BloomBilinearOracle(object):
    """BilinearNet whose user and/or item embedding layer is a BloomEmbedding
    (spotlight/layers.py:74-244); `user_bloom` / `item_bloom` are bloom_desc() dicts or None."""

    def __init__(self, user_emb, item_emb, user_bias, item_bias, user_bloom=None, item_bloom=None,
                 opt='adagrad', lr=1e-2, eps=None, betas=(0.9, 0.999), weight_decay=0.0, lr_decay=0.0, step=0):
        assert lib().slko_sizeof_bmodel() == C.sizeof(_BModel)
        f = lambda a: np.array(a, dtype=np.float32, order='C', copy=True)
        self.p = [f(user_emb), f(item_emb), f(user_bias).reshape(-1), f(item_bias).reshape(-1)]
        if eps is None:
            eps = 1e-10 if opt.startswith('adagrad') else 1e-8
        self.s1 = [np.zeros_like(p) for p in self.p]
        self.s2 = [np.zeros_like(p) for p in self.p]
        self.m = _BModel()
        for t in range(4):
            self.m.p[t] = self.p[t].ctypes.data
            self.m.s1[t] = self.s1[t].ctypes.data
            self.m.s2[t] = self.s2[t].ctypes.data
        self.m.num_users, self.m.num_items = self.p[2].size, self.p[3].size
        self.m.dim = self.p[0].shape[1]
        self.m.opt_kind = OPTS[opt]
        self.m.lr, self.m.eps = lr, eps
        self.m.beta1, self.m.beta2 = betas
        self.m.weight_decay, self.m.lr_decay = weight_decay, lr_decay
        self.m.step = step
        for side, desc in enumerate((user_bloom, item_bloom)):
            b = self.m.bloom[side]
            if desc is None:
                b.n_hash = 0
                assert self.p[side].shape[0] == (self.m.num_users, self.m.num_items)[side]
                continue
            b.rows, b.n_hash = self.p[side].shape[0], desc['n_hash']
            b.padding_idx, b.skip_row = desc['padding_idx'], desc['skip_row']
            for h, s in enumerate(desc['seeds']):
                b.seeds[h] = s

    @property
    def step_count(self):
        return int(self.m.step)

    def predict(self, users, items=None):
        users = np.ascontiguousarray(np.atleast_1d(users), dtype=np.int64)
        if items is None:
            n = int(self.m.num_items)
        else:
            items = np.ascontiguousarray(items, dtype=np.int64)
            n = items.size
        out = np.empty(n, dtype=np.float32)
        lib().slko_bloom_predict(C.byref(self.m), _ptr(users), C.c_int64(users.size), _ptr(items), C.c_int64(n),
                                 _ptr(out))
        return out

    def step(self, users, pos, neg, loss='bpr', n_neg=1, want_grads=False):
        users = np.ascontiguousarray(users, dtype=np.int64)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        neg = np.ascontiguousarray(neg, dtype=np.int64).ravel()
        loss_out = C.c_float()
        dg, dgp = None, None
        if want_grads:
            dg = [np.zeros_like(p) for p in self.p]
            dgp = (C.c_void_p * 4)(*[g.ctypes.data for g in dg])
        rc = lib().slko_bloom_step(C.byref(self.m), _ptr(users), _ptr(pos), _ptr(neg), C.c_int64(users.size),
                                   C.c_int(n_neg), C.c_int(LOSSES[loss]), C.byref(loss_out), dgp)
        assert rc == 0
        return (float(loss_out.value), dg) if want_grads else float(loss_out.value)

    def train(self, rng, users, items, batch_size, loss='bpr', n_neg=1, neg_in=None, want_negs=False):
        users = np.ascontiguousarray(users, dtype=np.int64)
        items = np.ascontiguousarray(items, dtype=np.int64)
        n = users.size
        nn = n_neg if loss == 'adaptive_hinge' else 1
        n_mb = (n + batch_size - 1) // batch_size
        mb_loss = np.empty(n_mb, dtype=np.float32)
        neg_out = np.empty(n * nn, dtype=np.int64) if want_negs else None
        if neg_in is not None:
            neg_in = np.ascontiguousarray(neg_in, dtype=np.int64).ravel()
        rc = lib().slko_bloom_train(C.byref(self.m), C.byref(rng._r) if rng is not None else None, _ptr(users),
                                    _ptr(items), C.c_int64(n), C.c_int64(batch_size), C.c_int(LOSSES[loss]),
                                    C.c_int(n_neg), _ptr(neg_in), _ptr(neg_out), _ptr(mb_loss))
        assert rc == 0
        return (mb_loss, neg_out) if want_negs else mb_loss
