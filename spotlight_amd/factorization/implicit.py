"""ImplicitFactorizationModel -- drop-in for spotlight/factorization/implicit.py:22-311.

Same constructor, fit(), predict(), error messages and random-state behaviour as the
reference; everything inside the epoch loop (negative sampling, both forward passes, loss,
backward, optimizer update) is one C-ABI call into csrc/libspotlight_hip.so per epoch.
"""
import time

import numpy as np
import torch
import torch.optim as optim

from spotlight_amd import _native
from spotlight_amd.factorization._components import _predict_process_ids
from spotlight_amd.factorization.representations import BilinearNet
from spotlight_amd.helpers import _repr_model
from spotlight_amd.layers import BloomEmbedding, ScaledEmbedding, ZeroEmbedding
from spotlight_amd.torch_utils import set_seed

_ENGINES = {}
# fit(): epochs of at most this many negative draws prepare the next epoch while training.  Larger epochs do not gain:
# at 10^8 interactions the two lanes time-slice the chip (train 71 ms + prepare 38 ms alone, 100 ms together:
# profiles/r02_m_fit_pipelined_at_1e8_no_gain.json)
_PIPELINE_MAX_DRAWS = 1 << 22
_DEFERRED_CHECK_MIN = 1 << 22  # ids per fit() from which the id-range checks run on worker threads beside the upload
_PREFETCH = True  # large epochs: the next epoch's first chunk is prepared beside the last passes of this one (test switch)
# Item tables of at least this many rows train with their biases and the biases' Adagrad accumulator interleaved for the duration
# of fit() (include/spotlight_hip.h: slk_bias_shadow_begin): on such tables a minibatch's biases share no cache line and the two
# scalars cost the item pass a quarter of its memory requests (12.5 % of them are saved: C5 shard).
_BIAS_SHADOW_MIN_ITEMS = 1 << 24
# Minibatches of at least this many interactions (pair losses, plain tables, row-sparse optimizers) train on a DOUBLED user table
# for the duration of fit() (include/spotlight_hip.h: slk_user_pingpong_begin): the user pass writes a user's updated row to the
# copy that does not hold the current one and no pre-step-row record (a row per interaction: 13 % of the pass's traffic), the item
# pass gathers the pre-step row where it still stands.  Below, the passes are latency-bound and the record is cache-resident.
_USER_PINGPONG_MIN_BATCH = 1 << 17


def _engine_for(device):
    """One slk_ctx per (process, HIP device)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _ENGINES:
        _ENGINES[index] = _native.Engine(index)
    return _ENGINES[index]


# What a fit() asks of the shared ctx FOR ITS OWN DURATION (it used to be set once, process-wide, on the
# ctx every model, evaluation and to_sequence of the process share).  Epochs are runs of multi-chunk training calls: the next
# chunk's negatives + sorts go to the ctx's second stream beside the current chunk's passes (include/spotlight_hip.h, option
# "overlap_prep"; a bare ctx keeps them in line).
_FIT_OPTIONS = {'overlap_prep': 1}


def fit_scope(model):
    """`with fit_scope(model):` -- _FIT_OPTIONS on the model's engine for the block, the previous values afterwards."""
    net = getattr(model, '_net', None)
    device = net.tables()[0].device if (net is not None and hasattr(net, 'tables')) else _model_device()
    return _engine_for(device).options(**_FIT_OPTIONS)


def _stream_for(device):
    """Raw hipStream_t of torch's current stream on `device`."""
    return torch.cuda.current_stream(device).cuda_stream


_PREP = {}


def _prep_lane_for(device):
    """(engine, raw stream) on which the NEXT epoch's shuffle and negatives are prepared while the current epoch trains
    (fit() of small datasets, see ImplicitFactorizationModel.fit): a second slk_ctx with its own scratch and a side HIP
    stream.  Under the GPU-less test harness it is a second ctx of the same (synchronous, single-threaded) emulator library:
    two ctxs, as on the GPU -- what the prep lane does to its RNG state must not reach the training ctx."""
    engine = _engine_for(device)
    if _native._LIB is None or engine._lib is not _native._LIB:
        twin = getattr(engine, '_prep_twin', None)
        if twin is None:
            twin = engine._prep_twin = _native.Engine(0, lib=engine._lib)
        return twin, _stream_for(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _PREP:
        _PREP[index] = (_native.Engine(index), torch.cuda.Stream(device))
    prep_engine, side = _PREP[index]
    return prep_engine, side.cuda_stream


def _model_device():
    """The HIP device models live on.  (The GPU-less test harness substitutes these three
    hooks to drive the same host logic through its emulator build of the kernels.)"""
    if not torch.cuda.is_available():
        raise RuntimeError('spotlight_amd needs a HIP device (MI355X): torch.cuda.is_available() '
                           'is False and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _state_tensor(state, key, like):
    if key not in state:
        state[key] = torch.zeros_like(like, memory_format=torch.preserve_format)
    return state[key]


def ids_to_device(ids, device):
    """int64 device tensor of a host id array: uploaded in the array's own dtype (int32 in the reference's
    datasets: half the PCIe bytes) and widened on the device, instead of widening on the host first."""
    ids = np.ascontiguousarray(ids)
    if ids.dtype not in (np.int32, np.int64):
        ids = ids.astype(np.int64)
    return torch.from_numpy(ids).to(device).to(torch.int64)


def device_epoch_shuffle(engine, random_state, n, d_perm, arrays, stream):
    """d_dst = d_src[numpy-exact shuffle of arange(n)] for every (d_src, d_dst, row_len) of `arrays`,
    drawn from the engine's RNG state (slk_shuffle_perm: the reference's `shuffle`, torch_utils.py:35-52).
    If the device shuffle reports a failure (it cannot, short of a 12-sigma rejection tail) the
    permutation is drawn by numpy on the host from `random_state`, as the reference does, and the
    engine's RNG state is re-synchronised -- the results are identical either way.
    `arrays` may be a callable returning the list: it is called AFTER the permutation has been drawn (the permutation
    needs only n, so an upload of the ids can still be in flight while it is computed: IdUpload)."""
    try:
        engine.shuffle_perm(n, d_perm.data_ptr(), stream=stream)
    except _native.SlkError:
        order = np.arange(n)
        random_state.shuffle(order)
        d_perm.copy_(torch.from_numpy(order))
        engine.rng_set_state(random_state.get_state())
    if callable(arrays):
        arrays = arrays()
    for entry in arrays:
        if len(entry) == 4:  # ('pairs', d_pairs, d_users_dst, d_items_dst): both id arrays from their packed form in one pass
            _, d_pairs, d_users_dst, d_items_dst = entry
            engine.gather_id_pairs(d_pairs.data_ptr(), d_perm.data_ptr(), n, d_users_dst.data_ptr(), d_items_dst.data_ptr(),
                                   stream=stream)
            continue
        d_src, d_dst, row_len = entry
        engine.gather_rows_i64(d_src.data_ptr(), d_perm.data_ptr(), n, row_len, d_dst.data_ptr(), stream=stream)


class IdUpload(object):
    """The host -> HBM copy of fit()'s id arrays on a worker thread (ids_to_device; the copies release the GIL), so that the
    first epoch's permutation -- which needs only len(ids) -- is drawn on the GPU meanwhile.  result() joins and returns the
    int64 device tensors; the copies run on the worker's (default) stream, and a pageable copy has completed when the call
    returns, so after the join the data is in place for any stream."""

    def __init__(self, arrays, device):
        import threading
        self._out = [None] * len(arrays)
        self._err = []

        def work():
            try:
                if device.type == 'cuda':
                    torch.cuda.set_device(device)
                for k, a in enumerate(arrays):
                    self._out[k] = ids_to_device(a, device)
                if device.type == 'cuda':
                    torch.cuda.current_stream(device).synchronize()  # the widening kernels on this thread's stream
            except BaseException as e:  # re-raised by result()
                self._err.append(e)
        self._thread = threading.Thread(target=work, name='spotlight-id-upload')
        self._thread.start()

    def result(self):
        self._thread.join()
        if self._err:
            raise self._err[0]
        return self._out


class _Background(object):
    """fn() on a worker thread (or at once); join() returns its value or re-raises what it raised.  (Module level on purpose: a
    class created inside fit() would form a reference cycle through its closure and keep the epoch's id buffers -- GBs at
    bench scale -- out of the caching allocator until the cyclic collector runs; the next fit() then paid 20-50 ms of
    hipMalloc, profiles/r04_t_fit_first_epoch_probe.txt.)"""

    def __init__(self, fn, threaded):
        self._fn, self._out, self._err, self._thread = fn, None, None, None
        if threaded:
            import threading
            self._thread = threading.Thread(target=self._run, name='spotlight-epoch-shuffle')
            self._thread.start()
        else:
            self._run()

    def _run(self):
        try:
            self._out = self._fn()
        except BaseException as e:  # noqa: BLE001 -- re-raised by join()
            self._err = e
        self._fn = None

    def join(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._err is not None:
            raise self._err
        return self._out


class _Deferred(object):
    """Every function of `fns` now, or each on a worker thread of its own; result() joins and re-raises the first error in
    the order of `fns` (the order the serial checks would have raised in)."""

    def __init__(self, fns, threaded):
        self.threaded = bool(threaded)
        self._errs, self._threads = [None] * len(fns), []

        def work(k):
            try:
                fns[k]()
            except BaseException as e:  # noqa: BLE001 -- re-raised by result()
                self._errs[k] = e
        for k in range(len(fns)):
            if self.threaded:
                import threading
                self._threads.append(threading.Thread(target=work, args=(k,), name='spotlight-id-check'))
                self._threads[-1].start()
            else:
                work(k)
                if self._errs[k] is not None:
                    break

    def join(self):
        for t in self._threads:
            t.join()
        self._threads = []

    def result(self):
        self.join()
        for k, e in enumerate(self._errs):
            if e is not None:
                self._errs = [None] * len(self._errs)
                raise e


def _reject_negative_ids(ids):
    """The reference's torch embedding raises IndexError on a negative id; the kernels address rows with
    unsigned 32-bit ids, so a negative one must never reach them."""
    id_min = ids if isinstance(ids, int) else np.asarray(ids).min()
    if id_min < 0:
        raise IndexError('index out of range in self')


class _OptimizerBinding(object):
    """Maps a torch.optim object onto slk_optim.  The torch optimizer stays the owner of
    hyper-parameters and state tensors (so state_dict / pickle / resuming fit() behave as
    in the reference); the kernels update those tensors in place."""

    def __init__(self, optimizer, params, sparse):
        self.optimizer = optimizer
        self.params = params
        groups = optimizer.param_groups
        if len(groups) != 1:
            raise NotImplementedError('spotlight_amd supports a single optimizer param group')
        g = groups[0]
        if g.get('maximize', False):
            raise NotImplementedError('maximize=True is not supported')
        st = optimizer.state
        if isinstance(optimizer, optim.Adagrad):
            wd = g['weight_decay']
            if sparse and wd != 0:
                # torch/optim/adagrad.py:355-358
                raise RuntimeError('weight_decay option is not compatible with sparse gradients')
            self.kind = 'adagrad_dense' if wd != 0 else 'adagrad'
            self.hp = dict(lr=g['lr'], eps=g['eps'], weight_decay=wd, lr_decay=g['lr_decay'])
            self.s1 = [st[p]['sum'] for p in params]
            self.s2 = None
        elif isinstance(optimizer, optim.SparseAdam):
            if not sparse:
                raise RuntimeError('SparseAdam does not support dense gradients, please consider '
                                   'Adam instead')
            self.kind = 'sparse_adam'
            self.hp = dict(lr=g['lr'], eps=g['eps'], betas=g['betas'])
            for p in params:
                st[p].setdefault('step', 0)
            self.s1 = [_state_tensor(st[p], 'exp_avg', p) for p in params]
            self.s2 = [_state_tensor(st[p], 'exp_avg_sq', p) for p in params]
        elif type(optimizer) is optim.Adam:
            if sparse:
                raise RuntimeError('Adam does not support sparse gradients, please consider '
                                   'SparseAdam instead')
            if g.get('amsgrad', False):
                raise NotImplementedError('amsgrad=True is not supported')
            self.kind = 'adam_dense'
            self.hp = dict(lr=g['lr'], eps=g['eps'], betas=g['betas'], weight_decay=g['weight_decay'])
            for p in params:
                if 'step' not in st[p]:
                    st[p]['step'] = torch.tensor(0.0, dtype=torch.float32)
            self.s1 = [_state_tensor(st[p], 'exp_avg', p) for p in params]
            self.s2 = [_state_tensor(st[p], 'exp_avg_sq', p) for p in params]
        elif type(optimizer) is optim.SGD:
            # torch/optim/sgd.py, the plain form: param.add_(grad, alpha=-lr).  Stateless, and a zero gradient is a zero update, so
            # dense and sparse gradients give the same row-sparse step.  Momentum / weight decay / Nesterov would touch every row
            # every step and have no fused form.
            if g.get('momentum', 0) != 0 or g.get('weight_decay', 0) != 0 or g.get('nesterov', False):
                raise NotImplementedError('torch.optim.SGD has a fused gfx950 update only with momentum=0, weight_decay=0, '
                                          'nesterov=False')
            self.kind = 'sgd'
            self.hp = dict(lr=g['lr'])
            self.s1 = None
            self.s2 = None
        else:
            raise NotImplementedError(
                'optimizer {} has no fused gfx950 update; supported: Adam (default), Adagrad, SparseAdam, SGD '
                '(momentum=0)'.format(type(optimizer).__name__))
        for t in (self.s1 or []) + (self.s2 or []):
            if not t.is_contiguous():
                raise RuntimeError('optimizer state must be contiguous')

    def steps_taken(self):
        if self.kind == 'sgd':  # stateless (and touching optimizer.state[...] would create an entry)
            return 0
        step = self.optimizer.state[self.params[0]].get('step', 0)
        return int(step.item()) if torch.is_tensor(step) else int(step)

    def as_struct(self):
        return _native.make_optim(self.kind, [t.data_ptr() for t in self.s1] if self.s1 else None,
                                  [t.data_ptr() for t in self.s2] if self.s2 else None,
                                  step=self.steps_taken(), **self.hp)

    def store_steps(self, step):
        if self.kind == 'sgd':  # stateless: torch keeps no step count for it either
            return
        for p in self.params:
            s = self.optimizer.state[p]
            if torch.is_tensor(s.get('step')):
                s['step'].fill_(float(step))
            else:
                s['step'] = int(step)


class ImplicitFactorizationModel(object):
    """Implicit-feedback matrix factorization trained by negative sampling.

    Parameters and semantics follow spotlight/factorization/implicit.py:76-88.  Two
    notes specific to this implementation:

    * `use_cuda` is accepted for signature compatibility; the model always lives on the HIP
      device (there is no CPU path).
    * `representation` may be a :class:`BilinearNet` (fused kernels) or any torch module with the reference's
      `forward(user_ids, item_ids)` contract; a custom module, or an `optimizer_func` whose optimizer has no fused update
      (anything but Adam / Adagrad / SparseAdam / plain SGD), trains through the AUTOGRAD ROUTE: the reference's own
      minibatch loop (implicit.py:208-252) in stock PyTorch-ROCm ops on the HIP device -- host numpy shuffle and
      negatives exactly as the reference draws them, the embedding gathers and their backward through this package's
      kernels when the module is built from this package's layers.  Slower than the fused path by the reference's own
      per-minibatch overheads, never on the CPU.
    """

    def __init__(self, loss='pointwise', embedding_dim=32, n_iter=10, batch_size=256, l2=0.0,
                 learning_rate=1e-2, optimizer_func=None, use_cuda=False, representation=None,
                 sparse=False, random_state=None, num_negative_samples=5):

        assert loss in ('pointwise', 'bpr', 'hinge', 'adaptive_hinge')

        self._loss = loss
        self._embedding_dim = embedding_dim
        self._n_iter = n_iter
        self._learning_rate = learning_rate
        self._batch_size = batch_size
        self._l2 = l2
        self._use_cuda = use_cuda
        self._representation = representation
        self._sparse = sparse
        self._optimizer_func = optimizer_func
        self._random_state = random_state or np.random.RandomState()
        self._num_negative_samples = num_negative_samples

        self._num_users = None
        self._num_items = None
        self._net = None
        self._optimizer = None
        self._loss_func = None
        self._binding = None
        self._autograd_route = False

        # consumes one draw of the stream, like the reference (implicit.py:114-115)
        set_seed(self._random_state.randint(-10**8, 10**8), cuda=self._use_cuda)

    def __repr__(self):
        return _repr_model(self)

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_binding'] = None  # holds raw pointers; rebuilt on the next fit()
        return state

    @property
    def _initialized(self):
        return self._net is not None

    def _initialize(self, interactions):
        self._num_users, self._num_items = interactions.num_users, interactions.num_items
        self._autograd_route = False
        if self._representation is not None:
            net = self._representation
        else:
            net = BilinearNet(self._num_users, self._num_items, self._embedding_dim,
                              sparse=self._sparse)
        if not isinstance(net, BilinearNet):
            self._autograd_route = True  # an arbitrary module: the reference's loop through autograd (see the class docstring)
            self._fused_ranks = None
            self._batch_scores = None    # (evaluation's whole-table fast path needs BilinearNet's tables; predict() serves it)
        else:
            for layer in (net.user_embeddings, net.item_embeddings):
                if not isinstance(layer, (ScaledEmbedding, BloomEmbedding)) and type(layer) is not torch.nn.Embedding:
                    self._autograd_route = True
            if not isinstance(net.user_biases, (ZeroEmbedding, torch.nn.Embedding)):
                self._autograd_route = True
        self._net = net.to(_model_device())

        if self._optimizer_func is None:
            self._optimizer = optim.Adam(self._net.parameters(), weight_decay=self._l2,
                                         lr=self._learning_rate)
        else:
            self._optimizer = self._optimizer_func(self._net.parameters())
        self._loss_func = self._loss  # the loss is fused into the kernel; kept for introspection
        self._binding = None
        if not self._autograd_route:
            try:
                self._bind()
            except NotImplementedError:
                # an optimizer without a fused update (RMSprop, SGD with momentum, ...): the reference accepts any
                # optimizer_func (implicit.py:150), so does this model -- through the autograd route
                self._autograd_route = True

    def _bind(self):
        if self._binding is None:
            tables = self._net.tables()
            for t in tables:
                if not (t.device.type == _model_device().type and t.is_contiguous()
                        and t.dtype == torch.float32):
                    raise RuntimeError('model tables must be contiguous fp32 tensors on the HIP device')
            self._binding = _OptimizerBinding(self._optimizer, tables, self._sparse)
        return self._binding

    def _check_user_id_max(self, user_ids):
        user_id_max = user_ids if isinstance(user_ids, int) else user_ids.max()
        if user_id_max >= self._num_users:
            raise ValueError('Maximum user id greater than number of users in model.')

    def _check_item_id_max(self, item_ids):
        item_id_max = item_ids if isinstance(item_ids, int) else item_ids.max()
        if item_id_max >= self._num_items:
            raise ValueError('Maximum item id greater than number of items in model.')

    def _id_checks(self, user_ids, item_ids):
        """The checks of _check_input as separate functions, in the order they raise (one reduction over a host array each)."""
        return [lambda: self._check_user_id_max(user_ids), lambda: _reject_negative_ids(user_ids),
                lambda: self._check_item_id_max(item_ids), lambda: _reject_negative_ids(item_ids)]

    def _check_input(self, user_ids, item_ids, allow_items_none=False):
        checks = self._id_checks(user_ids, item_ids)
        for f in (checks[:2] if allow_items_none and item_ids is None else checks):
            f()

    def _slk_tables(self):
        return self._net.slk_tables()

    def fit(self, interactions, verbose=False):
        """Fit the model; repeated calls resume from the current parameters and optimizer
        state (implicit.py:184-252)."""
        with fit_scope(self):
            return self._fit(interactions, verbose)

    def _fit(self, interactions, verbose):
        user_ids, item_ids = interactions.user_ids, interactions.item_ids

        if not self._initialized:
            self._initialize(interactions)

        n = len(user_ids)
        nn = self._num_negative_samples if self._loss == 'adaptive_hinge' else 1
        autograd = getattr(self, '_autograd_route', False)
        small = self._n_iter > 1 and n * nn <= _PIPELINE_MAX_DRAWS
        # the id-range checks (implicit.py:169-182): four reductions over the host arrays, 12 ms at 2^25 ids.  On the large-epoch
        # path they run on worker threads beside the id upload and the first epoch's shuffle; their verdict is collected before
        # the first training call is enqueued (nothing the caller can observe has changed by then: the RandomState is restored,
        # the tables are untouched).
        check = _Deferred(self._id_checks(user_ids, item_ids),
                          threaded=not (autograd or small) and n >= _DEFERRED_CHECK_MIN)
        if not check.threaded:
            check.result()
        # diagnostic: set model._fit_timeline = [] before fit() to collect (label, host time) marks of the large-epoch loop
        tl = getattr(self, '_fit_timeline', None)
        mark = (lambda label: tl.append((label, time.perf_counter()))) if tl is not None else (lambda label: None)
        mark('checks started')
        if autograd:
            return self._fit_autograd(user_ids, item_ids, verbose)
        binding = self._bind()
        device = self._net.tables()[0].device
        engine = _engine_for(device)
        stream = _stream_for(device)
        tables = self._slk_tables()
        # bpr / hinge leave the user biases' gradient at exactly zero (+g - g), so a model the reference initialised (ZeroEmbedding)
        # keeps them identically zero; checked HERE, on the device, once per fit(): the user pass then does not fetch a cache line
        # per interaction for a table of zeros (include/spotlight_hip.h: SLK_TABLES_USER_BIAS_ZERO).  Same values either way.
        if (self._loss in ('bpr', 'hinge') and binding.kind in ('adagrad', 'sgd') and not tables.user_bloom and not tables.item_bloom
                and not bool(self._net.tables()[2].any())):
            tables.flags |= _native.TABLES_USER_BIAS_ZERO
        n_minibatches = (n + self._batch_size - 1) // self._batch_size
        mb_loss = torch.empty(n_minibatches, dtype=torch.float32, device=device)

        engine.bilinear_reserve(tables, binding.as_struct(), n, self._batch_size, self._loss,
                                self._num_negative_samples, stream=stream)
        mark('scratch reserved')
        # ids go to the device once (on a worker thread: the first epoch's permutation is drawn meanwhile); every epoch's
        # permutation x[shuffle_indices] of them (torch_utils.py:35-52) is computed there, bit-exact with numpy's Fisher-Yates
        if small:
            d_users0, d_items0 = IdUpload([user_ids, item_ids], device).result()
            return self._fit_pipelined(binding, engine, device, stream, tables, d_users0, d_items0, n, nn, mb_loss, verbose)
        # Large epochs: a three-stage pipeline over epochs (round 4), every stage consuming the ONE MT19937 stream in the
        # reference's order -- shuffle(e), negatives(e), shuffle(e + 1), ... (implicit.py:212-221, torch_utils.py:46-47,
        # sampling.py:34):
        #   train lane   the passes of epoch e (slk_bilinear_train; it draws nothing: its negatives exist already)
        #   prep stream  the negatives of the WHOLE epoch e + 1 and the sorts of its first chunk (slk_bilinear_prefetch), drawn
        #                beside the last passes of epoch e as soon as epoch e's training call has been enqueued
        #   prep lane    the shuffle + id gathers of epoch e + 2 on the second slk_ctx / HIP stream, from the stream position
        #                behind epoch e + 1's negatives -- known an epoch ahead (slk_rng_get_state_sampled); on a worker thread,
        #                because the numpy-exact shuffle reads counts back and would hold the host up
        # so that in the steady state nothing but the passes is on the critical path.  Same ids, negatives and RandomState as
        # the serial loop, bit-identical tables (tests/test_host_model.py, tests/test_gpu_model.py).  Before: 28.9 ms per epoch
        # at 2^25 interactions, of which 3.3 ms shuffle and 2.1 ms first-chunk prep ran with no pass beside them
        # (profiles/r04_m_fit_epoch_breakdown.txt).
        prep, prep_stream = _prep_lane_for(device)
        threaded = _PREFETCH and prep is not engine and engine._lib is _native._LIB  # (the test harness's emulator is single-threaded)
        n_slots = min(self._n_iter, 3 if _PREFETCH else 2)
        bufs = [(torch.empty(n, dtype=torch.int64, device=device), torch.empty(n, dtype=torch.int64, device=device))
                for _ in range(max(n_slots, 1))]
        d_perm = torch.empty(n, dtype=torch.int64, device=device)
        d_pairs = unpacked = None
        # (the upload starts AFTER the epoch buffers exist: its worker allocates from the same caching allocator, and racing it
        # for the blocks the previous fit() left there sent this thread to hipMalloc, 15 ms per pair of buffers)
        mark('epoch buffers')
        upload = IdUpload([user_ids, item_ids], device)

        def shuffle_into(slot, state):
            """bufs[slot] = ids[numpy-exact shuffle] drawn from `state` on the prep lane; returns the state behind the shuffle
            (reading it waits for the prep lane's stream: the gathers are complete)."""
            def sources():
                # the uploaded ids, packed once per fit() into (user, item) pairs of 32 bits: a shuffled interaction is then
                # ONE random 8-byte read for both arrays (slk_gather_id_pairs)
                nonlocal d_pairs, unpacked
                if d_pairs is None:
                    unpacked = upload.result()  # (kept until the prep lane has been waited for, below)
                    upload._out = None
                    d_pairs = torch.empty(2 * n, dtype=torch.int32, device=device)
                    # INVARIANT: with the id checks on worker threads (`check.threaded`) these ids are NOT validated
                    # yet -- they are truncated to uint32, permuted and gathered as DATA only.  Nothing may index a table with
                    # them, and no training call or prefetch may be enqueued, before `check.result()` has returned
                    # (tests/test_host_model.py::test_id_checks_on_worker_threads_raise_like_the_serial_ones).
                    prep.pack_id_pairs(unpacked[0].data_ptr(), unpacked[1].data_ptr(), n, d_pairs.data_ptr(), stream=prep_stream)
                return [('pairs', d_pairs, bufs[slot][0], bufs[slot][1])]
            fallback = np.random.RandomState()  # (device_epoch_shuffle's host fall-back draws from it: private to this job)
            fallback.set_state(state)
            if device.type == 'cuda':
                torch.cuda.set_device(device)
            prep.rng_set_state(state)
            nonlocal unpacked
            device_epoch_shuffle(prep, fallback, n, d_perm, sources, prep_stream)
            after = prep.rng_get_state()
            unpacked = None  # the 64-bit copies of the ids: packed by now
            return after

        def _Job(slot, state):
            # shuffle_into on a worker thread (or at once, under the single-threaded test harness)
            return _Background(lambda: shuffle_into(slot, state), threaded)

        if self._n_iter <= 0:  # nothing to train: no shuffle is drawn, the RandomState stays where it is (as in the reference)
            upload.result()
            check.result()
            return
        mark('loop set up')
        # self._random_state runs AHEAD of training; `consumed` is the state the reference would hold at this point -- restored on
        # every way out, the normal one included
        consumed = self._random_state.get_state()
        job = None
        shadow = engine.bias_shadow(tables, binding.as_struct(), stream=stream,
                                    enabled=(binding.kind == 'adagrad' and self._num_items >= _BIAS_SHADOW_MIN_ITEMS and
                                             self._batch_size >= 4096 and not tables.item_bloom))
        shadowed = False
        pingpong = engine.user_pingpong(tables, binding.as_struct(), stream=stream,
                                        enabled=(self._loss in ('pointwise', 'bpr', 'hinge') and binding.kind in ('adagrad', 'sparse_adam', 'sgd')
                                                 and min(self._batch_size, n) >= _USER_PINGPONG_MIN_BATCH
                                                 and not tables.user_bloom and not tables.item_bloom))
        pingponged = False
        try:
            state = shuffle_into(0, consumed)  # the first epoch's shuffle: the one nothing hides
            mark('shuffle 0')
            check.result()  # raises what _check_input raised
            shadow.__enter__()  # (from here to the `finally` the item-bias tensors are stale: nothing below reads them)
            shadowed = True
            pingpong.__enter__()  # (... and the user-embedding tensor holds only some of the current rows)
            pingponged = True
            if pingpong.active:  # the scope's {dL/dscore, src} pairs are wider than the one-table layout's side array: grown now,
                # not inside the first training call
                engine.bilinear_reserve(tables, binding.as_struct(), n, self._batch_size, self._loss, self._num_negative_samples,
                                        stream=stream)
            if _PREFETCH:
                engine.bilinear_prefetch(tables, binding.as_struct(), bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), n, self._batch_size,
                                         self._loss, self._num_negative_samples, state=state, stream=stream)
                after_negs = engine.rng_get_state_sampled()  # behind negatives(0) -- or `state` itself if nothing was drawn ahead
                drawn_ahead = engine.get_stat('prefetch_pending') == 2
            else:
                engine.rng_set_state(state)
                after_negs, drawn_ahead = None, False
            if drawn_ahead and self._n_iter > 1:
                job = _Job(1 % n_slots, after_negs)  # shuffle(1) beside the passes of epoch 0
            mark('prefetch 0')
            for epoch_num in range(self._n_iter):
                d_users, d_items = bufs[epoch_num % n_slots]
                ostruct = binding.as_struct()
                engine.bilinear_train(tables, ostruct, d_users.data_ptr(), d_items.data_ptr(), n,
                                      self._batch_size, self._loss, self._num_negative_samples,
                                      mb_loss.data_ptr(), stream=stream)
                binding.store_steps(ostruct.step)
                mark('train %d enqueued' % epoch_num)
                # the state the reference holds at the end of this epoch: behind its negatives
                consumed = after_negs if drawn_ahead else engine.rng_get_state_sampled()
                if epoch_num + 1 < self._n_iter:
                    nxt = (epoch_num + 1) % n_slots
                    # shuffle(e + 1): prepared an epoch ago on the worker -- or now, beside the last passes of this epoch
                    state = job.join() if job is not None else shuffle_into(nxt, consumed)
                    job = None
                    mark('shuffle %d joined' % (epoch_num + 1))
                    if _PREFETCH:
                        engine.bilinear_prefetch(tables, binding.as_struct(), bufs[nxt][0].data_ptr(), bufs[nxt][1].data_ptr(), n,
                                                 self._batch_size, self._loss, self._num_negative_samples, state=state, stream=stream)
                        after_negs = engine.rng_get_state_sampled()
                        drawn_ahead = engine.get_stat('prefetch_pending') == 2
                        if drawn_ahead and epoch_num + 2 < self._n_iter:
                            job = _Job((epoch_num + 2) % n_slots, after_negs)
                        mark('prefetch %d' % (epoch_num + 1))
                    else:
                        engine.rng_set_state(state)

                # mean of per-minibatch loss.item() (implicit.py:240,245): one D2H per epoch; also waits for the epoch's kernels
                epoch_loss = float(mb_loss.double().mean().item())
                mark('epoch %d done' % epoch_num)
                engine.check()  # errors the training kernels can only report through the ctx

                if verbose:
                    print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

                if np.isnan(epoch_loss) or epoch_loss == 0.0:
                    # the reference stops here having consumed the stream up to this epoch's negatives only
                    raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))
        finally:
            try:
                try:
                    if pingponged:
                        pingpong.__exit__(None, None, None)  # the rows whose current copy is the ctx's back into torch's tensor
                finally:
                    if shadowed:
                        shadow.__exit__(None, None, None)  # the trained biases and their accumulator back into torch's tensors
            finally:  # (whatever the write-back raised, the threads are joined and the RandomState is the reference's)
                if job is not None:
                    try:
                        job.join()
                    except BaseException:  # noqa: BLE001 -- the exception already on its way out wins
                        pass
                upload.result()  # (joins the upload thread on the exceptional paths too)
                check.join()
                self._random_state.set_state(consumed)
                # a chunk prepared ahead for an epoch that will not run (an exception left the loop): its draws are dropped and the
                # ctx's stream is put back where the reference's would be, so that the next training call on this engine -- any
                # model's -- finds no stale prefetch (slk_bilinear_train would refuse it once)
                if _PREFETCH and engine.get_stat('prefetch_pending'):
                    engine.rng_set_state(consumed)
                # the epoch's device buffers go back to the caching allocator NOW (closures above hold cells, not tensors, once
                # these names are cleared): the next fit() reuses them instead of allocating
                del bufs[:]
                d_perm = d_pairs = unpacked = None
                upload._out = None

    def _fit_pipelined(self, binding, engine, device, stream, tables, d_users0, d_items0, n, nn, mb_loss, verbose):
        """The epoch loop for datasets of the reference's own scale (MovieLens-100K: 80 000 interactions per epoch): there an
        epoch trains in about a millisecond and the numpy-exact device shuffle -- a chain of small launches and read-backs --
        costs as much again.  The RandomState stream is consumed in the reference's order (shuffle of epoch e, negatives of
        epoch e, shuffle of epoch e + 1, ...), but nothing in training touches it, so epoch e + 1's shuffle and negatives are
        drawn on a second slk_ctx / HIP stream while epoch e trains from its own (already drawn) negatives.  Same ids, same
        negatives, same RandomState afterwards, bit-identical tables; only the wall time changes."""
        prep, prep_stream = _prep_lane_for(device)
        torch.cuda.current_stream(device).synchronize() if device.type == 'cuda' else None  # the id upload is complete
        bufs = [(torch.empty_like(d_users0), torch.empty_like(d_items0),
                 torch.empty(n * nn, dtype=torch.int64, device=device)) for _ in range(2)]
        d_perm = torch.empty(n, dtype=torch.int64, device=device)

        def prepare(slot):
            # shuffle, then the negatives: one MT19937 stream, consumed on the GPU exactly as numpy would consume it
            prep.rng_set_state(self._random_state.get_state())
            d_users, d_items, d_neg = bufs[slot]
            device_epoch_shuffle(prep, self._random_state, n, d_perm, [(d_users0, d_users, 1), (d_items0, d_items, 1)],
                                 prep_stream)
            # one randint per minibatch == one contiguous draw over the epoch (sampling.py:34)
            prep.sample_items(self._num_items, n * nn, d_neg.data_ptr(), stream=prep_stream)
            self._random_state.set_state(prep.rng_get_state())  # synchronises the prep stream

        prepare(0)
        for epoch_num in range(self._n_iter):
            d_users, d_items, d_neg = bufs[epoch_num % 2]
            ostruct = binding.as_struct()
            engine.bilinear_train(tables, ostruct, d_users.data_ptr(), d_items.data_ptr(), n, self._batch_size, self._loss,
                                  self._num_negative_samples, mb_loss.data_ptr(), d_neg_in=d_neg.data_ptr(), stream=stream)
            binding.store_steps(ostruct.step)
            state_after_epoch = self._random_state.get_state()
            if epoch_num + 1 < self._n_iter:
                prepare((epoch_num + 1) % 2)  # overlaps the training kernels of this epoch

            epoch_loss = float(mb_loss.double().mean().item())  # also waits for this epoch's kernels
            engine.check()  # errors the training kernels can only report through the ctx (e.g. an abandoned launch)

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                # the reference stops here having consumed the stream up to this epoch's negatives only
                self._random_state.set_state(state_after_epoch)
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _fit_autograd(self, user_ids, item_ids, verbose):
        """The reference's epoch / minibatch loop (implicit.py:208-252, 254-275) for models without a fused path: host numpy
        shuffle and negatives from `random_state` (the reference's exact consumption of the stream), forward / loss /
        backward / optimizer.step() as stock torch ops on the HIP device."""
        from spotlight_amd import losses as _losses
        from spotlight_amd.sampling import sample_items
        from spotlight_amd.torch_utils import minibatch, shuffle
        loss_func = {'pointwise': _losses.pointwise_loss, 'bpr': _losses.bpr_loss, 'hinge': _losses.hinge_loss,
                     'adaptive_hinge': _losses.adaptive_hinge_loss}[self._loss]
        device = next(self._net.parameters()).device
        self._net.train(True)

        def negatives_for(batch_user, n_draws):
            return torch.from_numpy(sample_items(self._num_items, n_draws, random_state=self._random_state)).to(device)

        for epoch_num in range(self._n_iter):
            users, items = shuffle(user_ids, item_ids, random_state=self._random_state)
            user_ids_tensor = torch.from_numpy(np.ascontiguousarray(users).astype(np.int64)).to(device)
            item_ids_tensor = torch.from_numpy(np.ascontiguousarray(items).astype(np.int64)).to(device)
            epoch_loss, n_mb = 0.0, 0
            for batch_user, batch_item in minibatch(user_ids_tensor, item_ids_tensor, batch_size=self._batch_size):
                positive_prediction = self._net(batch_user, batch_item)
                if self._loss == 'adaptive_hinge':
                    # implicit.py:266-275: users repeated user-major, ONE draw of B * n, scores viewed as [n, B]
                    n = self._num_negative_samples
                    batch_size = batch_user.size(0)
                    negative_items = negatives_for(batch_user, batch_size * n).view(batch_size * n)
                    batch_user_rep = batch_user.view(batch_size, 1).expand(batch_size, n).reshape(-1)
                    negative_prediction = self._net(batch_user_rep, negative_items).view(n, batch_size)
                else:
                    negative_prediction = self._net(batch_user, negatives_for(batch_user, len(batch_user)))
                self._optimizer.zero_grad()
                loss = loss_func(positive_prediction, negative_prediction)
                epoch_loss += loss.item()
                loss.backward()
                self._optimizer.step()
                n_mb += 1
            epoch_loss /= n_mb
            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))
            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def predict(self, user_ids, item_ids=None):
        """Scores for one user against all/some items, or for explicit (user, item) pairs;
        returns a flat np.float32 array (implicit.py:277-311)."""
        self._check_input(user_ids, item_ids, allow_items_none=True)
        self._net.train(False)

        if getattr(self, '_autograd_route', False) and not isinstance(self._net, BilinearNet):
            # a custom module scores with its own forward, as in the reference
            users, items, n = _predict_process_ids(user_ids, item_ids, self._num_items)
            device = next(self._net.parameters()).device
            d_users = torch.from_numpy(np.broadcast_to(users, (n,)).astype(np.int64)).to(device)
            d_items = torch.from_numpy(items if items is not None else np.arange(self._num_items, dtype=np.int64)).to(device)
            with torch.no_grad():
                return self._net(d_users, d_items).cpu().numpy().flatten()

        users, items, n = _predict_process_ids(user_ids, item_ids, self._num_items)
        device = self._net.tables()[0].device
        engine = _engine_for(device)
        d_users = torch.from_numpy(users).to(device)
        d_items = torch.from_numpy(items).to(device) if items is not None else None
        out = torch.empty(n, dtype=torch.float32, device=device)
        engine.bilinear_predict(self._slk_tables(), d_users.data_ptr(), users.size,
                                d_items.data_ptr() if d_items is not None else None, n,
                                out.data_ptr(), _stream_for(device))
        return out.cpu().numpy().flatten()

    def _fused_ranks(self, user_ids, row_group, row_target, exc_off, exc_items):
        """Average ranks of the rows' target items (evaluation.mrr_score's fast path): one counting sweep of the item table
        per 64 rows on the matrix cores, no score matrix (csrc/slk_eval.hip, slk_bilinear_rank)."""
        users = np.ascontiguousarray(np.asarray(user_ids).reshape(-1), dtype=np.int64)
        self._check_input(users, None, allow_items_none=True)
        self._net.train(False)
        device = self._net.tables()[0].device
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(np.int64))).to(device)
        d_users, d_rg, d_rt = dev(users), dev(row_group), dev(row_target)
        d_eo, d_ei = (dev(exc_off), dev(exc_items)) if exc_off is not None else (None, None)
        ranks = torch.empty(len(row_group), dtype=torch.float64, device=device)
        _engine_for(device).bilinear_rank(self._slk_tables(), d_users.data_ptr(), users.size, d_rg.data_ptr(), d_rt.data_ptr(),
                                          len(row_group), d_eo.data_ptr() if d_eo is not None else None,
                                          d_ei.data_ptr() if d_ei is not None else None, ranks.data_ptr(), _stream_for(device))
        return ranks.cpu().numpy()

    def _batch_scores(self, user_ids):
        """[len(user_ids), num_items] device tensor: row r == predict(user_ids[r]) (bit-identical), a
        tile of users per pass over the item table (csrc/slk_eval.hip); used by evaluation.mrr_score."""
        users = np.ascontiguousarray(np.asarray(user_ids).reshape(-1), dtype=np.int64)
        self._check_input(users, None, allow_items_none=True)
        self._net.train(False)
        device = self._net.tables()[0].device
        d_users = torch.from_numpy(users).to(device)
        out = torch.empty((users.size, self._num_items), dtype=torch.float32, device=device)
        _engine_for(device).bilinear_scores(self._slk_tables(), d_users.data_ptr(), users.size, out.data_ptr(),
                                            _stream_for(device))
        return out
