"""ExplicitFactorizationModel -- drop-in for spotlight/factorization/explicit.py:21-284.

The explicit-feedback sibling of ImplicitFactorizationModel: the same BilinearNet and the same
fused gather / dot / backward / row-update kernels, with observed ratings instead of sampled
negatives and the regression / poisson / logistic losses of spotlight/losses.py:169-244
(include/spotlight_hip.h: slk_bilinear_train_explicit).  Same constructor, fit(), predict(),
error behaviour and random-state consumption as the reference (one draw in the constructor, one
numpy-exact shuffle of the three arrays per epoch, computed on the device).
"""
import numpy as np
import torch

from spotlight_amd.factorization._components import _predict_process_ids
from spotlight_amd.factorization import implicit as _host
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel
from spotlight_amd.torch_utils import set_seed


class ExplicitFactorizationModel(ImplicitFactorizationModel):
    """Explicit-feedback matrix factorization (ratings).  Parameters follow
    spotlight/factorization/explicit.py:68-79; `use_cuda` is accepted for signature compatibility
    (the model always lives on the HIP device), `representation` may be a BilinearNet."""

    def __init__(self, loss='regression', embedding_dim=32, n_iter=10, batch_size=256, l2=0.0,
                 learning_rate=1e-2, optimizer_func=None, use_cuda=False, representation=None, sparse=False,
                 random_state=None):

        assert loss in ('regression', 'poisson', 'logistic')

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

        self._num_users = None
        self._num_items = None
        self._net = None
        self._optimizer = None
        self._loss_func = None
        self._binding = None

        # consumes one draw of the stream, like the reference (explicit.py:103-104)
        set_seed(self._random_state.randint(-10**8, 10**8), cuda=self._use_cuda)

    def fit(self, interactions, verbose=False):
        """Fit the model on interactions that carry ratings; repeated calls resume
        (explicit.py:173-243)."""
        with _host.fit_scope(self):
            return self._fit(interactions, verbose)

    def _fit(self, interactions, verbose):
        user_ids, item_ids = interactions.user_ids, interactions.item_ids

        if not self._initialized:
            self._initialize(interactions)

        self._check_input(user_ids, item_ids)
        if interactions.ratings is None:
            # the reference fails inside shuffle() when it indexes None (explicit.py:201-204)
            raise TypeError("'NoneType' object is not subscriptable: explicit feedback needs interactions.ratings")

        binding = self._bind()
        device = self._net.tables()[0].device
        engine = _host._engine_for(device)
        stream = _host._stream_for(device)
        tables = self._slk_tables()
        n = len(user_ids)
        n_minibatches = (n + self._batch_size - 1) // self._batch_size
        mb_loss = torch.empty(n_minibatches, dtype=torch.float32, device=device)

        # ids and ratings go to the device once; the ratings ride through the shuffle as int64 bit
        # patterns (slk_gather_rows_i64 moves 8-byte elements)
        ratings = np.ascontiguousarray(interactions.ratings, dtype=np.float32)
        d_users0 = _host.ids_to_device(user_ids, device)
        d_items0 = _host.ids_to_device(item_ids, device)
        d_ratings0 = torch.from_numpy(ratings).to(device).view(torch.int32).to(torch.int64)
        if self._n_iter > 1 and n <= _host._PIPELINE_MAX_DRAWS:
            return self._fit_pipelined(binding, engine, device, stream, tables, d_users0, d_items0, d_ratings0, n, mb_loss,
                                       verbose)
        d_users, d_items, d_rbits = (torch.empty_like(d_users0), torch.empty_like(d_items0),
                                     torch.empty_like(d_ratings0))
        d_perm = torch.empty(n, dtype=torch.int64, device=device)
        for epoch_num in range(self._n_iter):
            engine.rng_set_state(self._random_state.get_state())
            _host.device_epoch_shuffle(engine, self._random_state, n, d_perm,
                                       [(d_users0, d_users, 1), (d_items0, d_items, 1), (d_ratings0, d_rbits, 1)],
                                       stream)
            d_ratings = d_rbits.to(torch.int32).view(torch.float32)
            ostruct = binding.as_struct()
            engine.bilinear_train_explicit(tables, ostruct, d_users.data_ptr(), d_items.data_ptr(),
                                           d_ratings.data_ptr(), n, self._batch_size, self._loss, mb_loss.data_ptr(),
                                           stream=stream)
            binding.store_steps(ostruct.step)
            self._random_state.set_state(engine.rng_get_state())  # synchronises the stream

            epoch_loss = float(mb_loss.double().mean().item())

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _fit_pipelined(self, binding, engine, device, stream, tables, d_users0, d_items0, d_ratings0, n, mb_loss, verbose):
        """The epoch loop for datasets of the reference's own scale (its README example fits MovieLens-100K with batch_size
        256): training draws nothing from the RandomState, so epoch e + 1's shuffle is computed on a second slk_ctx / HIP
        stream while epoch e trains (see ImplicitFactorizationModel._fit_pipelined).  Same permutations, same RandomState
        afterwards, bit-identical tables."""
        prep, prep_stream = _host._prep_lane_for(device)
        torch.cuda.current_stream(device).synchronize() if device.type == 'cuda' else None  # the uploads are complete
        bufs = [(torch.empty_like(d_users0), torch.empty_like(d_items0), torch.empty_like(d_ratings0),
                 torch.empty(n, dtype=torch.float32, device=device)) for _ in range(2)]
        d_perm = torch.empty(n, dtype=torch.int64, device=device)
        side = torch.cuda.ExternalStream(prep_stream, device=device) if device.type == 'cuda' else None

        def prepare(slot):
            prep.rng_set_state(self._random_state.get_state())
            d_users, d_items, d_rbits, d_ratings = bufs[slot]
            _host.device_epoch_shuffle(prep, self._random_state, n, d_perm,
                                       [(d_users0, d_users, 1), (d_items0, d_items, 1), (d_ratings0, d_rbits, 1)], prep_stream)
            if side is not None:
                with torch.cuda.stream(side):
                    d_ratings.copy_(d_rbits.to(torch.int32).view(torch.float32))
            else:
                d_ratings.copy_(d_rbits.to(torch.int32).view(torch.float32))
            self._random_state.set_state(prep.rng_get_state())  # synchronises the prep stream (the conversion included)

        prepare(0)
        for epoch_num in range(self._n_iter):
            d_users, d_items, _, d_ratings = bufs[epoch_num % 2]
            ostruct = binding.as_struct()
            engine.bilinear_train_explicit(tables, ostruct, d_users.data_ptr(), d_items.data_ptr(), d_ratings.data_ptr(), n,
                                           self._batch_size, self._loss, mb_loss.data_ptr(), stream=stream)
            binding.store_steps(ostruct.step)
            state_after_epoch = self._random_state.get_state()
            if epoch_num + 1 < self._n_iter:
                prepare((epoch_num + 1) % 2)  # overlaps the training kernels of this epoch

            epoch_loss = float(mb_loss.double().mean().item())  # also waits for this epoch's kernels
            engine.check()

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                self._random_state.set_state(state_after_epoch)
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def predict(self, user_ids, item_ids=None):
        """Predicted ratings for one user against all/some items, or for explicit (user, item) pairs:
        exp() of the score under the poisson loss, sigmoid() under the logistic one (explicit.py:245-284)."""
        self._check_input(user_ids, item_ids, allow_items_none=True)
        self._net.train(False)

        users, items, n = _predict_process_ids(user_ids, item_ids, self._num_items)
        device = self._net.tables()[0].device
        engine = _host._engine_for(device)
        d_users = torch.from_numpy(users).to(device)
        d_items = torch.from_numpy(items).to(device) if items is not None else None
        out = torch.empty(n, dtype=torch.float32, device=device)
        engine.bilinear_predict(self._slk_tables(), d_users.data_ptr(), users.size,
                                d_items.data_ptr() if d_items is not None else None, n,
                                out.data_ptr(), _host._stream_for(device))
        if self._loss == 'poisson':
            out = torch.exp(out)
        elif self._loss == 'logistic':
            out = torch.sigmoid(out)
        return out.cpu().numpy().flatten()

    # evaluation.mrr_score's device fast path ranks raw scores; predicted ratings (exp / sigmoid of them)
    # take the generic predict() route, as in the reference (evaluation.py:9-56)
    _batch_scores = None
    _fused_ranks = None
