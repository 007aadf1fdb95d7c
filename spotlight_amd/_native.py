"""ctypes binding of include/spotlight_hip.h (libspotlight_hip.so, gfx950).

There is deliberately no fallback: if the HIP library is missing or fails to load, every
entry point of spotlight_amd that needs it raises.  `bind()` is split out so that the test
harness can bind the same prototypes onto its emulator build of the same sources
(tests/emu) -- spotlight_amd itself only ever loads csrc/libspotlight_hip.so.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libspotlight_hip.so')

SLK_ABI_VERSION = 12
SLK_OK, SLK_EIO, SLK_ENOMEM, SLK_EINVAL, SLK_ERANGE = 0, -5, -12, -22, -34

LOSS_KINDS = {'pointwise': 0, 'bpr': 1, 'hinge': 2, 'adaptive_hinge': 3,
              'regression': 4, 'poisson': 5, 'logistic': 6}
OPT_KINDS = {'adagrad': 0, 'sparse_adam': 1, 'adam_dense': 2, 'adagrad_dense': 3, 'sgd': 4}
KERNEL_CLASSES = {'sample': 0, 'prep': 1, 'user_pass': 2, 'item_pass': 3, 'dense_sweep': 4, 'score': 5,
                  'exchange': 6, 'seq_pass': 7, 'epoch': 8}


class SlkBloom(C.Structure):
    _fields_ = [('rows', C.c_int64), ('n_hash', C.c_int32), ('reserved', C.c_int32),
                ('padding_idx', C.c_int64), ('skip_row', C.c_int64), ('seeds', C.c_uint32 * 8)]


class SlkTables(C.Structure):
    _fields_ = [('d_param', C.c_void_p * 4), ('num_users', C.c_int64), ('num_items', C.c_int64),
                ('dim', C.c_int32), ('flags', C.c_int32),
                ('user_bloom', C.POINTER(SlkBloom)), ('item_bloom', C.POINTER(SlkBloom))]


class SlkOptim(C.Structure):
    _fields_ = [('kind', C.c_int32), ('reserved', C.c_int32), ('step', C.c_int64),
                ('lr', C.c_double), ('eps', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('weight_decay', C.c_double), ('lr_decay', C.c_double),
                ('d_state1', C.c_void_p * 4), ('d_state2', C.c_void_p * 4)]


class SlkShard(C.Structure):
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('num_items_global', C.c_int64),
                ('global_batch', C.c_int64)]


_PROTOTYPES = {
    'slk_abi_version': (C.c_int, []),
    'slk_ctx_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    'slk_ctx_destroy': (None, [C.c_void_p]),
    'slk_last_error': (C.c_char_p, [C.c_void_p]),
    'slk_ctx_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    'slk_ctx_get_option': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    'slk_bias_shadow_begin': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_void_p]),
    'slk_bias_shadow_end': (C.c_int, [C.c_void_p, C.c_void_p]),
    'slk_bias_shadow_abort': (C.c_int, [C.c_void_p]),
    'slk_user_pingpong_begin': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_void_p]),
    'slk_user_pingpong_end': (C.c_int, [C.c_void_p, C.c_void_p]),
    'slk_user_pingpong_abort': (C.c_int, [C.c_void_p]),
    'slk_ctx_get_stat': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    'slk_rng_set_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'slk_rng_get_state': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    'slk_rng_get_state_sampled': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    'slk_sample_items': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_bilinear_train': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_bilinear_train_explicit': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    'slk_bilinear_prefetch': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'slk_bilinear_reserve': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int64, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    'slk_bilinear_predict': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_poolnet_train': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_poolnet_reserve': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int32, C.c_int32, C.c_void_p]),
    'slk_poolnet_predict': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_shuffle_perm': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_gather_rows_i64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_pack_id_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_gather_id_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_to_sequence_plan': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_void_p]),
    'slk_to_sequence_fill': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_embedding_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p]),
    'slk_embedding_backward_plan': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                              C.POINTER(C.c_int64), C.c_void_p]),
    'slk_embedding_backward_fill': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_bilinear_scores': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_poolnet_scores': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_void_p]),
    'slk_bilinear_rank': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_poolnet_rank': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_rank_targets': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'slk_shard_buffer_floats': (C.c_int64, [C.c_int32, C.c_int64]),
    'slk_shard_reserve': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkShard), C.c_int64, C.c_int64]),
    'slk_shard_chunk_begin': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkShard), C.c_void_p, C.c_void_p,
                                        C.c_int64, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_shard_chunk_commit': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkShard), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    'slk_shard_gather': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_int32, C.c_void_p, C.c_void_p]),
    'slk_shard_user_pass': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.POINTER(SlkShard),
                                      C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_void_p]),
    'slk_shard_item_pass': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    'slk_shard_chunk_begin_adaptive': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkShard), C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_shard_score_pass': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_shard_adaptive_select': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p]),
    'slk_shard_user_pass_adaptive': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    'slk_profile_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'slk_profile_read': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'slk_profile_reset': (C.c_int, [C.c_void_p]),
    'slk_probe_stream': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                   C.POINTER(C.c_double), C.c_void_p]),
    'slk_probe_random_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                        C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    'slk_probe_sort': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    'slk_probe_step_ceiling': (C.c_int, [C.c_void_p, C.POINTER(SlkTables), C.POINTER(SlkOptim), C.c_int64, C.c_int32,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOTYPES))


def bind(cdll):
    """Attach the prototypes of include/spotlight_hip.h to a loaded library."""
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(cdll, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if cdll.slk_abi_version() != SLK_ABI_VERSION:
        raise ImportError('libspotlight_hip ABI %d != expected %d'
                          % (cdll.slk_abi_version(), SLK_ABI_VERSION))
    return cdll


_LIB = None


def load():
    """Loads csrc/libspotlight_hip.so; raises ImportError (never falls back) if it is absent."""
    global _LIB
    if _LIB is None:
        path = os.environ.get('SPOTLIGHT_HIP_LIB', LIB_PATH)  # another build of the same ABI (A/B measurements)
        if path != LIB_PATH:
            _LIB = bind(C.CDLL(path))
            return _LIB
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: build it with `python -m spotlight_amd.build` '
                              '(hipcc --offload-arch=gfx950).  spotlight_amd has no CPU fallback.'
                              % LIB_PATH)
        _LIB = bind(C.CDLL(LIB_PATH))
    return _LIB


class SlkError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, 'libspotlight_hip error %d: %s' % (code, text))
        self.code = code


class Engine(object):
    """One slk_ctx.  All pointer arguments are raw integer addresses of device memory
    (torch `tensor.data_ptr()`), `stream` is a raw hipStream_t (0 = default stream)."""

    def __init__(self, device_id=0, lib=None):
        self._lib = lib if lib is not None else load()
        ctx = C.c_void_p()
        rc = self._lib.slk_ctx_create(C.byref(ctx), int(device_id))
        if rc != SLK_OK:
            raise SlkError(rc, (self._lib.slk_last_error(None) or b'').decode())
        self._ctx = ctx
        self.device_id = int(device_id)
        # A/B measurements: SPOTLIGHT_HIP_OPTIONS="name=value,..." applies slk_ctx_set_option to every new ctx
        for kv in filter(None, os.environ.get('SPOTLIGHT_HIP_OPTIONS', '').split(',')):
            name, value = kv.split('=')
            self.set_option(name.strip(), int(value))

    def close(self):
        twin = getattr(self, '_prep_twin', None)  # the test harness's second ctx (factorization/implicit.py::_prep_lane_for)
        if twin is not None:
            self._prep_twin = None
            twin.close()
        if getattr(self, '_ctx', None):
            self._lib.slk_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """include/spotlight_hip.h: slk_ctx_set_option (tuning knobs; results do not change)."""
        self._check(self._lib.slk_ctx_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name):
        """include/spotlight_hip.h: slk_ctx_get_option (the current value of a tuning knob)."""
        v = C.c_int64(0)
        self._check(self._lib.slk_ctx_get_option(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def options(self, **values):
        """Context manager: the given options for the duration of the block, the previous values afterwards -- on every way out.
        A ctx is shared by every model of a process on its device (factorization/implicit.py::_engine_for), so a caller that
        wants a route for ITS work sets it for that work only:

            with engine.options(overlap_prep=1):
                ... training calls ...
        """
        engine = self

        class _Scoped(object):
            def __enter__(self_):
                self_.saved = [(k, engine.get_option(k)) for k in values]
                for k, v in values.items():
                    engine.set_option(k, v)
                return engine

            def __exit__(self_, *exc):
                for k, v in reversed(self_.saved):
                    engine.set_option(k, v)
                return False
        return _Scoped()

    def bias_shadow(self, tables, optim, stream=0, enabled=True):
        """include/spotlight_hip.h: slk_bias_shadow_begin / _end as a context manager -- the item biases and their Adagrad
        accumulator interleaved for the duration of the block (the caller's tensors are stale inside it and rewritten on every
        way out).  `enabled=False`: a no-op scope."""
        engine = self

        class _Shadow(object):
            active = False

            def __enter__(self_):
                if enabled:
                    rc = engine._lib.slk_bias_shadow_begin(engine._ctx, C.byref(tables), C.byref(optim), C.c_void_p(stream))
                    if rc == SLK_ENOMEM:
                        # 8 bytes per item row did not fit (0.8 GB at 10^8 rows): the scope is an optimisation, never a
                        # requirement -- training goes on in the two-array layout, bit-identically
                        return engine
                    engine._check(rc)
                    self_.active = True
                return engine

            def __exit__(self_, *exc):
                if self_.active:
                    self_.active = False
                    engine._check(engine._lib.slk_bias_shadow_end(engine._ctx, C.c_void_p(stream)))
                return False

            def abort(self_):
                """Close the scope without writing back (the caller's arrays are gone): slk_bias_shadow_abort."""
                if self_.active:
                    self_.active = False
                    engine._check(engine._lib.slk_bias_shadow_abort(engine._ctx))
        return _Shadow()

    def user_pingpong(self, tables, optim, stream=0, enabled=True):
        """include/spotlight_hip.h: slk_user_pingpong_begin / _end as a context manager -- the user embedding table doubled for
        the duration of the block: the user pass writes updated rows to the other copy, the item pass gathers pre-step rows
        where they still stand, no record is written (the caller's tensor is a mix of rows inside the block and made whole on
        every way out).  `enabled=False`: a no-op scope."""
        engine = self

        class _PingPong(object):
            active = False

            def __enter__(self_):
                if enabled:
                    rc = engine._lib.slk_user_pingpong_begin(engine._ctx, C.byref(tables), C.byref(optim), C.c_void_p(stream))
                    if rc == SLK_ENOMEM:  # the second copy did not fit: the scope is an optimisation, never a requirement
                        return engine
                    engine._check(rc)
                    self_.active = True
                return engine

            def __exit__(self_, *exc):
                if self_.active:
                    self_.active = False
                    engine._check(engine._lib.slk_user_pingpong_end(engine._ctx, C.c_void_p(stream)))
                return False

            def abort(self_):
                """Close the scope without writing back (the caller's arrays are gone): slk_user_pingpong_abort."""
                if self_.active:
                    self_.active = False
                    engine._check(engine._lib.slk_user_pingpong_abort(engine._ctx))
        return _PingPong()

    def get_stat(self, name):
        """include/spotlight_hip.h: slk_ctx_get_stat (diagnostics of the last calls)."""
        v = C.c_int64(0)
        self._check(self._lib.slk_ctx_get_stat(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def _check(self, rc):
        if rc != SLK_OK:
            raise SlkError(rc, (self._lib.slk_last_error(self._ctx) or b'').decode())

    # -- numpy RandomState hand-over ---------------------------------------------------
    def rng_set_state(self, state):
        """`state` = numpy RandomState.get_state() tuple."""
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        assert key.shape == (624,)
        self._check(self._lib.slk_rng_set_state(self._ctx, key.ctypes.data, int(state[2])))

    def rng_get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        self._check(self._lib.slk_rng_get_state(self._ctx, key.ctypes.data, C.byref(pos)))
        return ('MT19937', key, int(pos.value), 0, 0.0)

    def rng_get_state_sampled(self):
        """The stream position after the last draw of negatives this ctx enqueued, without waiting for the passes that consume
        them (include/spotlight_hip.h: slk_rng_get_state_sampled)."""
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        self._check(self._lib.slk_rng_get_state_sampled(self._ctx, key.ctypes.data, C.byref(pos)))
        return ('MT19937', key, int(pos.value), 0, 0.0)

    def check(self):
        """Synchronises the ctx's last stream and raises if a kernel reported a failure through the ctx (the sticky flags
        slk_rng_get_state reports: sampler ran dry, persistent epoch kernel abandoned a launch)."""
        self.rng_get_state()

    def sample_items(self, num_items, count, d_out, stream=0):
        self._check(self._lib.slk_sample_items(self._ctx, int(num_items), int(count), d_out, stream))

    # -- training / prediction ---------------------------------------------------------
    def bilinear_train(self, tables, optim, d_users, d_items, n, batch_size, loss, n_neg,
                       d_mb_loss, d_neg_in=None, d_neg_out=None, stream=0):
        self._check(self._lib.slk_bilinear_train(
            self._ctx, C.byref(tables), C.byref(optim), d_users, d_items, int(n), int(batch_size),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), int(n_neg), d_neg_in, d_neg_out,
            d_mb_loss, stream))

    def bilinear_train_explicit(self, tables, optim, d_users, d_items, d_ratings, n, batch_size, loss, d_mb_loss,
                                stream=0):
        self._check(self._lib.slk_bilinear_train_explicit(
            self._ctx, C.byref(tables), C.byref(optim), d_users, d_items, d_ratings, int(n), int(batch_size),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), d_mb_loss, stream))

    def bilinear_prefetch(self, tables, optim, d_users, d_items, n, batch_size, loss, n_neg, state=None, stream=0):
        """Prepares the first chunk of the next bilinear_train call with these arguments on the ctx's second stream; `state`: the
        numpy RandomState state its draws start from (written without waiting for the ctx's stream)."""
        key, pos = None, 0
        if state is not None:
            key = np.ascontiguousarray(state[1], dtype=np.uint32)
            pos = int(state[2])
        self._check(self._lib.slk_bilinear_prefetch(
            self._ctx, C.byref(tables), C.byref(optim), d_users, d_items, int(n), int(batch_size),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), int(n_neg),
            key.ctypes.data_as(C.c_void_p) if key is not None else None, pos, stream))

    def bilinear_reserve(self, tables, optim, n, batch_size, loss, n_neg, stream=0):
        self._check(self._lib.slk_bilinear_reserve(
            self._ctx, C.byref(tables), C.byref(optim), int(n), int(batch_size),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), int(n_neg), stream))

    def bilinear_predict(self, tables, d_users, n_users, d_items, n, d_out, stream=0):
        self._check(self._lib.slk_bilinear_predict(self._ctx, C.byref(tables), d_users, int(n_users),
                                                   d_items, int(n), d_out, stream))

    # -- PoolNet (sequence model) -------------------------------------------------------
    def poolnet_train(self, tables, optim, padding_idx, d_sequences, n_seq, seq_len, batch_size, loss, n_neg,
                      d_mb_loss, d_neg_in=None, d_neg_out=None, stream=0):
        self._check(self._lib.slk_poolnet_train(
            self._ctx, C.byref(tables), C.byref(optim), -1 if padding_idx is None else int(padding_idx),
            d_sequences, int(n_seq), int(seq_len), int(batch_size),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), int(n_neg), d_neg_in, d_neg_out,
            d_mb_loss, stream))

    def poolnet_reserve(self, tables, optim, n_seq, seq_len, batch_size, loss, n_neg, stream=0):
        self._check(self._lib.slk_poolnet_reserve(self._ctx, C.byref(tables), C.byref(optim), int(n_seq), int(seq_len),
                                                  int(batch_size), LOSS_KINDS[loss] if isinstance(loss, str) else int(loss),
                                                  int(n_neg), stream))

    def poolnet_predict(self, tables, d_sequence, seq_len, d_items, n, d_out, stream=0):
        self._check(self._lib.slk_poolnet_predict(self._ctx, C.byref(tables), d_sequence, int(seq_len), d_items,
                                                  int(n), d_out, stream))

    # -- epoch shuffle on the device (include/spotlight_hip.h: slk_shuffle_perm) ----------
    def shuffle_perm(self, n, d_perm_out, stream=0):
        self._check(self._lib.slk_shuffle_perm(self._ctx, int(n), d_perm_out, stream))

    def gather_rows_i64(self, d_src, d_perm, n, row_len, d_dst, stream=0):
        self._check(self._lib.slk_gather_rows_i64(self._ctx, d_src, d_perm, int(n), int(row_len), d_dst, stream))

    def pack_id_pairs(self, d_users, d_items, n, d_pairs, stream=0):
        """d_pairs[2 r], d_pairs[2 r + 1] = (uint32) d_users[r], d_items[r] (include/spotlight_hip.h: slk_pack_id_pairs)."""
        self._check(self._lib.slk_pack_id_pairs(self._ctx, d_users, d_items, int(n), d_pairs, stream))

    def gather_id_pairs(self, d_pairs, d_perm, n, d_users_out, d_items_out, stream=0):
        """Both id arrays of fit() through one permutation (slk_gather_id_pairs)."""
        self._check(self._lib.slk_gather_id_pairs(self._ctx, d_pairs, d_perm, int(n), d_users_out, d_items_out, stream))

    # -- Interactions.to_sequence on the device (include/spotlight_hip.h: slk_to_sequence_*) --
    def to_sequence_plan(self, d_users, d_items, d_timestamps, ts_kind, n, num_users, max_sequence_length,
                         step_size, min_length, stream=0):
        """Sorts and counts; returns the number of sequences the following to_sequence_fill writes."""
        rows = C.c_int64(0)
        self._check(self._lib.slk_to_sequence_plan(self._ctx, d_users, d_items, d_timestamps, int(ts_kind), int(n),
                                                   int(num_users), int(max_sequence_length), int(step_size),
                                                   int(min_length), C.byref(rows), stream))
        return int(rows.value)

    def to_sequence_fill(self, d_sequences, d_sequence_users, stream=0):
        self._check(self._lib.slk_to_sequence_fill(self._ctx, d_sequences, d_sequence_users, stream))

    # -- embedding front-end for the torch-side encoders (include/spotlight_hip.h: slk_embedding_*) --
    def embedding_forward(self, d_weight, rows, dim, bloom, d_ids, n, d_out, stream=0):
        self._check(self._lib.slk_embedding_forward(self._ctx, d_weight, int(rows), int(dim),
                                                    C.byref(bloom) if bloom is not None else None, d_ids, int(n),
                                                    d_out, stream))

    def embedding_backward_plan(self, rows, dim, bloom, padding_idx, d_ids, n, count_rows=False, stream=0):
        """Sorts the lookups by table row; with count_rows returns the number of distinct rows that
        receive a gradient (the size of the COO output of embedding_backward_fill)."""
        out = C.c_int64(0)
        self._check(self._lib.slk_embedding_backward_plan(
            self._ctx, int(rows), int(dim), C.byref(bloom) if bloom is not None else None,
            -1 if padding_idx is None else int(padding_idx), d_ids, int(n), C.byref(out) if count_rows else None, stream))
        return int(out.value)

    def embedding_backward_fill(self, d_grad_out, d_grad_dense=None, d_rows_out=None, d_values_out=None, stream=0):
        self._check(self._lib.slk_embedding_backward_fill(self._ctx, d_grad_out, d_grad_dense, d_rows_out,
                                                          d_values_out, stream))

    # -- evaluation: batched predict + on-GPU ranking (include/spotlight_hip.h) -----------
    def bilinear_scores(self, tables, d_users, n_users, d_out, stream=0):
        self._check(self._lib.slk_bilinear_scores(self._ctx, C.byref(tables), d_users, int(n_users), d_out, stream))

    def poolnet_scores(self, tables, d_sequences, n_seq, seq_len, d_out, stream=0):
        self._check(self._lib.slk_poolnet_scores(self._ctx, C.byref(tables), d_sequences, int(n_seq), int(seq_len),
                                                 d_out, stream))

    def bilinear_rank(self, tables, d_group_users, n_groups, d_row_group, d_row_target, n_rows, d_exc_off, d_exc_items,
                      d_rank_out, stream=0):
        self._check(self._lib.slk_bilinear_rank(self._ctx, C.byref(tables), d_group_users, int(n_groups), d_row_group,
                                                d_row_target, int(n_rows), d_exc_off, d_exc_items, d_rank_out, stream))

    def poolnet_rank(self, tables, d_group_sequences, n_groups, seq_len, d_row_group, d_row_target, n_rows, d_exc_off,
                     d_exc_items, d_rank_out, stream=0):
        self._check(self._lib.slk_poolnet_rank(self._ctx, C.byref(tables), d_group_sequences, int(n_groups), int(seq_len),
                                               d_row_group, d_row_target, int(n_rows), d_exc_off, d_exc_items, d_rank_out,
                                               stream))

    def rank_targets(self, d_scores, n_rows, num_items, d_exc_rows, d_exc_items, n_exc, d_tgt_rows, d_tgt_items,
                     n_tgt, d_rank_out, stream=0):
        self._check(self._lib.slk_rank_targets(self._ctx, d_scores, int(n_rows), int(num_items), d_exc_rows,
                                               d_exc_items, int(n_exc), d_tgt_rows, d_tgt_items, int(n_tgt),
                                               d_rank_out, stream))

    # -- row-sharded training phases (include/spotlight_hip.h: slk_shard_*) --------------
    def shard_buffer_floats(self, dim, slots):
        return int(self._lib.slk_shard_buffer_floats(int(dim), int(slots)))

    def shard_reserve(self, tables, shard, n, n_recv):
        self._check(self._lib.slk_shard_reserve(self._ctx, C.byref(tables), C.byref(shard), int(n), int(n_recv)))

    def shard_chunk_begin(self, tables, shard, d_users_local, d_items, n, mb_off, n_slices, d_send_ids,
                          d_send_counts, d_neg_in=None, d_neg_out=None, stream=0):
        m = len(mb_off) - 1
        off = (C.c_int64 * (m + 1))(*[int(x) for x in mb_off])
        self._check(self._lib.slk_shard_chunk_begin(self._ctx, C.byref(tables), C.byref(shard), d_users_local, d_items,
                                                    int(n), off, m, int(n_slices), d_neg_in, d_neg_out, d_send_ids,
                                                    d_send_counts, stream))

    def shard_chunk_commit(self, tables, shard, send_counts, recv_counts, d_recv_ids, stream=0):
        """send_counts / recv_counts: flat host sequences [world][units]."""
        sc = (C.c_int64 * len(send_counts))(*send_counts)
        rc = (C.c_int64 * len(recv_counts))(*recv_counts)
        self._check(self._lib.slk_shard_chunk_commit(self._ctx, C.byref(tables), C.byref(shard), sc, rc, d_recv_ids,
                                                     stream))

    def shard_gather(self, tables, unit, d_rows_out, stream=0):
        self._check(self._lib.slk_shard_gather(self._ctx, C.byref(tables), int(unit), d_rows_out, stream))

    def shard_user_pass(self, tables, optim, shard, unit, global_batch, loss, d_rows_in, d_grad_out, d_loss_out,
                        accumulate=False, stream=0):
        self._check(self._lib.slk_shard_user_pass(
            self._ctx, C.byref(tables), C.byref(optim), C.byref(shard), int(unit), int(global_batch),
            LOSS_KINDS[loss] if isinstance(loss, str) else int(loss), d_rows_in, d_grad_out, d_loss_out,
            1 if accumulate else 0, stream))

    def shard_chunk_begin_adaptive(self, tables, shard, d_users_local, d_items, n, mb_off, n_slices, n_neg, d_mb_pos,
                                   d_send_ids, d_send_counts, d_neg_in=None, d_neg_out=None, stream=0):
        m = len(mb_off) - 1
        off = (C.c_int64 * (m + 1))(*[int(x) for x in mb_off])
        self._check(self._lib.slk_shard_chunk_begin_adaptive(self._ctx, C.byref(tables), C.byref(shard), d_users_local, d_items,
                                                             int(n), off, m, int(n_slices), int(n_neg), d_neg_in, d_neg_out,
                                                             d_mb_pos, d_send_ids, d_send_counts, stream))

    def shard_score_pass(self, tables, unit, d_rows_in, d_scores, stream=0):
        self._check(self._lib.slk_shard_score_pass(self._ctx, C.byref(tables), int(unit), d_rows_in, d_scores, stream))

    def shard_adaptive_select(self, global_batch, n_neg, d_scores, d_gk, d_loss_out, report_loss, stream=0):
        self._check(self._lib.slk_shard_adaptive_select(self._ctx, int(global_batch), int(n_neg), d_scores, d_gk, d_loss_out,
                                                        1 if report_loss else 0, stream))

    def shard_user_pass_adaptive(self, tables, optim, unit, d_gk, d_rows_in, d_grad_out, stream=0):
        self._check(self._lib.slk_shard_user_pass_adaptive(self._ctx, C.byref(tables), C.byref(optim), int(unit), d_gk,
                                                           d_rows_in, d_grad_out, stream))

    def shard_item_pass(self, tables, optim, minibatch, d_grad_in, stream=0):
        self._check(self._lib.slk_shard_item_pass(self._ctx, C.byref(tables), C.byref(optim), int(minibatch),
                                                  d_grad_in, stream))

    # -- measurement -------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._lib.slk_profile_enable(self._ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.slk_profile_reset(self._ctx))

    def probe_stream(self, kind, d_a, d_b, d_c, n_floats, iters=10, stream=0):
        """Average ms of one float4 copy (kind 0) / triad (kind 1) launch over n_floats."""
        ms = C.c_double()
        self._check(self._lib.slk_probe_stream(self._ctx, int(kind), d_a, d_b, d_c, int(n_floats), int(iters),
                                               C.byref(ms), stream))
        return float(ms.value)

    def probe_random_rows(self, d_buf, rows, dim, layout, order, rmw, n_access, iters=10, stream=0):
        """Average ms of n_access row (+ state row) reads / read-modify-writes; see include/spotlight_hip.h."""
        ms = C.c_double()
        self._check(self._lib.slk_probe_random_rows(self._ctx, d_buf, int(rows), int(dim), int(layout), int(order), int(rmw),
                                                    int(n_access), int(iters), C.byref(ms), stream))
        return float(ms.value)

    def probe_sort(self, kind, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, bits, seg_len=0, iters=0, stream=0):
        """The engine's stable radix sort on caller-owned device arrays (kind 0: u32 + u32, 1: u32 + u64, 2: u64 + u32);
        returns the average ms of `iters` repeats (0.0 when iters == 0)."""
        ms = C.c_double()
        self._check(self._lib.slk_probe_sort(self._ctx, int(kind), d_keys_in, d_keys_out, d_vals_in, d_vals_out, int(n),
                                             int(seg_len), int(bits), int(iters), C.byref(ms), stream))
        return float(ms.value)

    def probe_step_ceiling(self, tables, optim, batch, iters=10, stream=0):
        """(user-side ms, item-side ms, distinct items touched) of the step's algorithmic accesses only."""
        um, im, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self._lib.slk_probe_step_ceiling(self._ctx, C.byref(tables), C.byref(optim), int(batch), int(iters),
                                                     C.byref(um), C.byref(im), C.byref(n), stream))
        return float(um.value), float(im.value), int(n.value)

    def profile_read(self):
        out = {}
        for name, cls in KERNEL_CLASSES.items():
            n, ms = C.c_int64(), C.c_double()
            self._check(self._lib.slk_profile_read(self._ctx, cls, C.byref(n), C.byref(ms)))
            out[name] = (int(n.value), float(ms.value))
        return out


BLOOM_SEEDS = (179424941, 179425457, 179425907, 179426369,
               179424977, 179425517, 179425943, 179426407)  # spotlight/layers.py:13-20, in order


def make_bloom(rows, n_hash, padding_idx=0, skip_row=0, seeds=None):
    b = SlkBloom()
    b.rows, b.n_hash = int(rows), int(n_hash)
    b.padding_idx = -1 if padding_idx is None else int(padding_idx)
    b.skip_row = -1 if skip_row is None else int(skip_row)
    for h, s in enumerate(seeds if seeds is not None else BLOOM_SEEDS[:n_hash]):
        b.seeds[h] = int(s)
    return b


TABLES_USER_BIAS_ZERO = 1  # include/spotlight_hip.h: SLK_TABLES_USER_BIAS_ZERO


def make_tables(ptrs, num_users, num_items, dim, user_bloom=None, item_bloom=None, user_bias_zero=False):
    """`user_bloom` / `item_bloom`: SlkBloom descriptors (kept alive by the returned struct).  `user_bias_zero`: the caller has
    CHECKED that the user-bias table is identically zero (slk_tables::flags, SLK_TABLES_USER_BIAS_ZERO)."""
    t = SlkTables()
    t.flags = TABLES_USER_BIAS_ZERO if user_bias_zero else 0
    for i in range(4):
        t.d_param[i] = ptrs[i]
    t.num_users, t.num_items, t.dim = int(num_users), int(num_items), int(dim)
    t._keep = (user_bloom, item_bloom)
    if user_bloom is not None:
        t.user_bloom = C.pointer(user_bloom)
    if item_bloom is not None:
        t.item_bloom = C.pointer(item_bloom)
    return t


def make_seq_tables(item_emb_ptr, item_bias_ptr, num_items, dim, item_bloom=None):
    """slk_tables of a PoolNet: slots 1 (item embeddings) and 3 (item biases); `item_bloom`: SlkBloom
    descriptor when the embedding layer is a BloomEmbedding (kept alive by the returned struct)."""
    t = SlkTables()
    t.d_param[1], t.d_param[3] = item_emb_ptr, item_bias_ptr
    t.num_users, t.num_items, t.dim = 0, int(num_items), int(dim)
    t._keep = (None, item_bloom)
    if item_bloom is not None:
        t.item_bloom = C.pointer(item_bloom)
    return t


def make_shard(world, rank, num_items_global, global_batch=0):
    sh = SlkShard()
    sh.world, sh.rank = int(world), int(rank)
    sh.num_items_global, sh.global_batch = int(num_items_global), int(global_batch)
    return sh


def make_optim(kind, state1, state2=None, lr=1e-2, eps=None, betas=(0.9, 0.999), weight_decay=0.0,
               lr_decay=0.0, step=0):
    o = SlkOptim()
    o.kind = OPT_KINDS[kind] if isinstance(kind, str) else int(kind)
    if eps is None:
        eps = 1e-10 if o.kind in (0, 3) else 1e-8
    o.step = int(step)
    o.lr, o.eps, o.beta1, o.beta2 = float(lr), float(eps), float(betas[0]), float(betas[1])
    o.weight_decay, o.lr_decay = float(weight_decay), float(lr_decay)
    for i in range(4):
        o.d_state1[i] = state1[i] if state1 is not None else None
        o.d_state2[i] = state2[i] if state2 is not None else None
    return o
