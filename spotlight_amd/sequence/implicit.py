"""ImplicitSequenceModel -- drop-in for spotlight/sequence/implicit.py:23-340.

Same constructor, fit(), predict(), error behaviour and random-state consumption as the
reference.  With the 'pooling' (PoolNet) representation everything inside the epoch loop
(negative sampling, PoolNet forward for the sequence and the negatives, masked loss, backward,
optimizer update) is one C-ABI call into csrc/libspotlight_hip.so per epoch
(include/spotlight_hip.h: slk_poolnet_train).  With 'lstm' / 'cnn' / 'mixture' (or any module
with the same two methods) the encoder body trains through torch autograd on MIOpen, fed by this
package's embedding front-end (spotlight_amd/embedding.py, csrc/slk_embed.hip); the epoch shuffle
and the negatives still come from the on-device numpy-exact MT19937 stream.
"""
import numpy as np
import torch
import torch.optim as optim

from spotlight_amd import _native
from spotlight_amd.factorization import implicit as _host
from spotlight_amd.factorization.implicit import _OptimizerBinding
from spotlight_amd.helpers import _repr_model
from spotlight_amd.layers import BloomEmbedding
from spotlight_amd.losses import adaptive_hinge_loss, bpr_loss, hinge_loss, pointwise_loss
from spotlight_amd.sequence.representations import PADDING_IDX, CNNNet, LSTMNet, MixtureLSTMNet, PoolNet
from spotlight_amd.torch_utils import set_seed

_LOSS_FUNCTIONS = {'pointwise': pointwise_loss, 'bpr': bpr_loss, 'hinge': hinge_loss,
                   'adaptive_hinge': adaptive_hinge_loss}


class _SeqOptimizerBinding(_OptimizerBinding):
    """slk_optim over PoolNet's two tables: ABI slots 1 (item embeddings) and 3 (item biases)."""

    def as_struct(self):
        slot = lambda xs: [None, xs[0].data_ptr(), None, xs[1].data_ptr()]
        return _native.make_optim(self.kind, slot(self.s1) if self.s1 else None, slot(self.s2) if self.s2 else None,
                                  step=self.steps_taken(), **self.hp)


class ImplicitSequenceModel(object):
    """Implicit-feedback sequence model (next-item prediction from the items seen so far).

    Parameters follow spotlight/sequence/implicit.py:85-97.  `representation`: 'pooling' or a
    :class:`PoolNet` (fused kernels end to end; item_embedding_layer may be a BloomEmbedding), or
    'cnn' / 'lstm' / 'mixture' / a module with user_representation + forward (encoder body on
    torch autograd, embedding lookups and their backward on this package's kernels).  `use_cuda`
    is accepted for signature compatibility; the model always lives on the HIP device.
    """

    def __init__(self, loss='pointwise', representation='pooling', embedding_dim=32, n_iter=10,
                 batch_size=256, l2=0.0, learning_rate=1e-2, optimizer_func=None, use_cuda=False,
                 sparse=False, random_state=None, num_negative_samples=5):

        assert loss in ('pointwise', 'bpr', 'hinge', 'adaptive_hinge')

        if isinstance(representation, str):
            assert representation in ('pooling', 'cnn', 'lstm', 'mixture')

        self._loss = loss
        self._representation = representation
        self._embedding_dim = embedding_dim
        self._n_iter = n_iter
        self._learning_rate = learning_rate
        self._batch_size = batch_size
        self._l2 = l2
        self._use_cuda = use_cuda
        self._sparse = sparse
        self._optimizer_func = optimizer_func
        self._random_state = random_state or np.random.RandomState()
        self._num_negative_samples = num_negative_samples

        self._num_items = None
        self._net = None
        self._optimizer = None
        self._loss_func = None
        self._binding = None

        # consumes one draw of the stream, like the reference (sequence/implicit.py:131-132)
        set_seed(self._random_state.randint(-10**8, 10**8), cuda=self._use_cuda)

    def __repr__(self):
        return _repr_model(self)

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_binding'] = None  # raw pointers; rebuilt on the next fit()
        return state

    @property
    def _initialized(self):
        return self._net is not None

    def _initialize(self, interactions):
        self._num_items = interactions.num_items
        if self._representation == 'pooling':
            net = PoolNet(self._num_items, self._embedding_dim, sparse=self._sparse)
        elif isinstance(self._representation, PoolNet):
            net = self._representation
        elif self._representation == 'cnn':
            net = CNNNet(self._num_items, self._embedding_dim, sparse=self._sparse)
        elif self._representation == 'lstm':
            net = LSTMNet(self._num_items, self._embedding_dim, sparse=self._sparse)
        elif self._representation == 'mixture':
            net = MixtureLSTMNet(self._num_items, self._embedding_dim, sparse=self._sparse)
        else:
            net = self._representation
        self._net = net.to(_host._model_device())

        if self._optimizer_func is None:
            self._optimizer = optim.Adam(self._net.parameters(), weight_decay=self._l2,
                                         lr=self._learning_rate)
        else:
            self._optimizer = self._optimizer_func(self._net.parameters())
        # PoolNet: fused into the kernel, the name is kept for introspection; other encoders call it
        self._loss_func = self._loss if isinstance(net, PoolNet) else _LOSS_FUNCTIONS[self._loss]
        self._binding = None

    def _bind(self):
        if self._binding is None:
            tables = self._net.tables()
            for t in tables:
                if not (t.device.type == _host._model_device().type and t.is_contiguous()
                        and t.dtype == torch.float32):
                    raise RuntimeError('model tables must be contiguous fp32 tensors on the HIP device')
            self._binding = _SeqOptimizerBinding(self._optimizer, tables, self._sparse)
        return self._binding

    def _check_input(self, item_ids):
        if isinstance(item_ids, int):
            item_id_max = item_ids
        else:
            item_id_max = item_ids.max()
        if item_id_max >= self._num_items:
            raise ValueError('Maximum item id greater than number of items in model.')
        _host._reject_negative_ids(item_ids)

    def _slk_tables(self):
        w = self._net.tables()
        layer = self._net.item_embeddings
        bloom = layer.descriptor() if isinstance(layer, BloomEmbedding) else None
        return _native.make_seq_tables(w[0].data_ptr(), w[1].data_ptr(), w[1].shape[0], w[0].shape[1],
                                       item_bloom=bloom)

    def _padding_idx(self):
        return self._net.item_embeddings.padding_idx

    def fit(self, interactions, verbose=False):
        """Fit the model on a SequenceInteractions dataset; repeated calls resume
        (sequence/implicit.py:193-264)."""
        with _host.fit_scope(self):
            return self._fit(interactions, verbose)

    def _fit(self, interactions, verbose):
        sequences = interactions.sequences

        if not self._initialized:
            self._initialize(interactions)

        self._check_input(sequences)

        if not isinstance(self._net, PoolNet):
            return self._fit_autograd(sequences, verbose)

        binding = self._bind()
        device = self._net.tables()[0].device
        engine = _host._engine_for(device)
        stream = _host._stream_for(device)
        tables = self._slk_tables()
        n_seq, seq_len = sequences.shape
        n_minibatches = (n_seq + self._batch_size - 1) // self._batch_size
        mb_loss = torch.empty(n_minibatches, dtype=torch.float32, device=device)

        engine.poolnet_reserve(tables, binding.as_struct(), n_seq, seq_len, self._batch_size, self._loss,
                               self._num_negative_samples, stream=stream)
        # the sequences go to the device once; `sequences` is rebound to its shuffled copy every epoch,
        # so successive epochs' permutations compose exactly as in the reference (:215-216) -- here by
        # gathering from the previous epoch's device array with a numpy-exact device permutation
        d_prev = _host.ids_to_device(sequences, device)
        d_sequences = torch.empty_like(d_prev)
        d_perm = torch.empty(n_seq, dtype=torch.int64, device=device)
        nn = self._num_negative_samples if self._loss == 'adaptive_hinge' else 1
        if self._n_iter > 1 and n_seq * seq_len * nn <= _host._PIPELINE_MAX_DRAWS:
            return self._fit_pipelined(binding, engine, device, stream, tables, d_prev, d_sequences, d_perm, n_seq, seq_len, nn,
                                       mb_loss, verbose)
        for epoch_num in range(self._n_iter):
            engine.rng_set_state(self._random_state.get_state())
            _host.device_epoch_shuffle(engine, self._random_state, n_seq, d_perm, [(d_prev, d_sequences, seq_len)],
                                       stream)
            ostruct = binding.as_struct()
            engine.poolnet_train(tables, ostruct, self._padding_idx(), d_sequences.data_ptr(), n_seq, seq_len,
                                 self._batch_size, self._loss, self._num_negative_samples,
                                 mb_loss.data_ptr(), stream=stream)
            d_prev, d_sequences = d_sequences, d_prev
            binding.store_steps(ostruct.step)
            self._random_state.set_state(engine.rng_get_state())  # synchronises the stream

            epoch_loss = float(mb_loss.double().mean().item())

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _fit_pipelined(self, binding, engine, device, stream, tables, d_prev, d_sequences, d_perm, n_seq, seq_len, nn, mb_loss,
                       verbose):
        """Small datasets (see ImplicitFactorizationModel._fit_pipelined): epoch e + 1's shuffle (composed with epoch e's, as
        the reference's rebinding of `sequences` does) and negatives are drawn on a second slk_ctx / HIP stream while epoch e
        trains from its own, already drawn negatives.  Same sequences, negatives, RandomState and tables, bit for bit."""
        prep, prep_stream = _host._prep_lane_for(device)
        torch.cuda.current_stream(device).synchronize() if device.type == 'cuda' else None  # the upload is complete
        negs = [torch.empty(n_seq * seq_len * nn, dtype=torch.int64, device=device) for _ in range(2)]

        def prepare(src, dst, d_neg):
            prep.rng_set_state(self._random_state.get_state())
            _host.device_epoch_shuffle(prep, self._random_state, n_seq, d_perm, [(src, dst, seq_len)], prep_stream)
            prep.sample_items(self._num_items, n_seq * seq_len * nn, d_neg.data_ptr(), stream=prep_stream)
            self._random_state.set_state(prep.rng_get_state())  # synchronises the prep stream

        prepare(d_prev, d_sequences, negs[0])
        for epoch_num in range(self._n_iter):
            ostruct = binding.as_struct()
            engine.poolnet_train(tables, ostruct, self._padding_idx(), d_sequences.data_ptr(), n_seq, seq_len,
                                 self._batch_size, self._loss, self._num_negative_samples, mb_loss.data_ptr(),
                                 d_neg_in=negs[epoch_num % 2].data_ptr(), stream=stream)
            binding.store_steps(ostruct.step)
            state_after_epoch = self._random_state.get_state()
            if epoch_num + 1 < self._n_iter:
                # this epoch's (shuffled) sequences are the source of the next permutation; its previous source is free
                prepare(d_sequences, d_prev, negs[(epoch_num + 1) % 2])
            d_prev, d_sequences = d_sequences, d_prev

            epoch_loss = float(mb_loss.double().mean().item())  # also waits for this epoch's kernels
            engine.check()

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                self._random_state.set_state(state_after_epoch)
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _fit_autograd(self, sequences, verbose):
        """The reference's epoch loop (sequence/implicit.py:213-264) for encoders whose body is a torch
        module: forward of the encoder, positive and negative scores, masked loss, backward and
        optimizer step are torch autograd; the embedding lookups inside (and their backward) are this
        package's kernels, and the shuffle + the negatives (`sample_items`, :266-276) come from the
        device MT19937 stream, bit-identical to numpy's under the same RandomState."""
        device = next(self._net.parameters()).device
        engine = _host._engine_for(device)
        stream = _host._stream_for(device)
        n_seq, seq_len = sequences.shape
        d_prev = _host.ids_to_device(sequences, device)
        d_sequences = torch.empty_like(d_prev)
        d_perm = torch.empty(n_seq, dtype=torch.int64, device=device)
        n_neg = self._num_negative_samples if self._loss == 'adaptive_hinge' else 1
        d_negatives = torch.empty(n_neg * self._batch_size * seq_len, dtype=torch.int64, device=device)
        self._net.train(True)
        for epoch_num in range(self._n_iter):
            engine.rng_set_state(self._random_state.get_state())
            _host.device_epoch_shuffle(engine, self._random_state, n_seq, d_perm, [(d_prev, d_sequences, seq_len)],
                                       stream)
            losses = []
            for lo in range(0, n_seq, self._batch_size):
                batch = d_sequences[lo:lo + self._batch_size]
                rows = batch.shape[0]
                representation, _ = self._net.user_representation(batch)
                positive = self._net(representation, batch)
                count = n_neg * rows * seq_len
                engine.sample_items(self._num_items, count, d_negatives.data_ptr(), stream=stream)
                negative_items = d_negatives[:count].view(n_neg * rows, seq_len)
                if self._loss == 'adaptive_hinge':
                    tiled = representation.repeat(*((n_neg,) + (1,) * (representation.dim() - 1)))
                    negative = self._net(tiled, negative_items).view(n_neg, rows, seq_len)
                else:
                    negative = self._net(representation, negative_items)
                self._optimizer.zero_grad()
                loss = self._loss_func(positive, negative, mask=(batch != PADDING_IDX))
                losses.append(loss.detach())
                loss.backward()
                self._optimizer.step()
            d_prev, d_sequences = d_sequences, d_prev
            self._random_state.set_state(engine.rng_get_state())  # synchronises the stream
            epoch_loss = float(torch.stack(losses).double().mean().item())

            if verbose:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))

            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _predict_autograd_free(self, sequences, item_ids):
        """predict() for the torch-side encoders (sequence/implicit.py:322-340)."""
        device = next(self._net.parameters()).device
        with torch.no_grad():
            sequence_var = torch.from_numpy(sequences.astype(np.int64).reshape(1, -1)).to(device)
            item_var = torch.from_numpy(np.asarray(item_ids).astype(np.int64)).to(device)
            _, final = self._net.user_representation(sequence_var)
            size = (len(item_var),) + final.size()[1:]
            out = self._net(final.expand(*size), item_var)
        return out.cpu().numpy().flatten()

    def predict(self, sequences, item_ids=None):
        """Scores of the next item after `sequences` (one sequence) for all items or for
        `item_ids`; flat np.float32 array (sequence/implicit.py:288-340)."""
        self._net.train(False)

        sequences = np.atleast_2d(sequences)

        all_items = item_ids is None
        if item_ids is None:
            item_ids = np.arange(self._num_items).reshape(-1, 1)

        self._check_input(item_ids)
        self._check_input(sequences)

        if not isinstance(self._net, PoolNet):
            return self._predict_autograd_free(sequences, item_ids)

        seq = np.ascontiguousarray(sequences.astype(np.int64).reshape(-1))
        items = np.ascontiguousarray(np.asarray(item_ids).astype(np.int64).reshape(-1))

        device = self._net.tables()[0].device
        engine = _host._engine_for(device)
        d_seq = torch.from_numpy(seq).to(device)
        d_items = None if all_items else torch.from_numpy(items).to(device)  # None: every item, in id order
        out = torch.empty(items.size, dtype=torch.float32, device=device)
        engine.poolnet_predict(self._slk_tables(), d_seq.data_ptr(), seq.size, d_items.data_ptr() if d_items is not None else None,
                               items.size, out.data_ptr(), _host._stream_for(device))
        return out.cpu().numpy().flatten()

    def _fused_ranks(self, sequences, row_group, row_target, exc_off, exc_items):
        """Average ranks of the rows' target items (evaluation.sequence_mrr_score's fast path): one counting sweep of the item
        table per 64 rows on the matrix cores, no score matrix (csrc/slk_eval.hip, slk_poolnet_rank).  None when the
        representation is not this package's PoolNet (the caller then ranks predict()'s rows)."""
        if not isinstance(self._net, PoolNet):
            return None
        self._net.train(False)
        sequences = np.atleast_2d(sequences)
        self._check_input(sequences)
        device = self._net.tables()[0].device
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(np.int64))).to(device)
        d_seqs, d_rg, d_rt = dev(sequences), dev(row_group), dev(row_target)
        d_eo, d_ei = (dev(exc_off), dev(exc_items)) if exc_off is not None else (None, None)
        ranks = torch.empty(len(row_group), dtype=torch.float64, device=device)
        _host._engine_for(device).poolnet_rank(self._slk_tables(), d_seqs.data_ptr(), sequences.shape[0], sequences.shape[1],
                                               d_rg.data_ptr(), d_rt.data_ptr(), len(row_group),
                                               d_eo.data_ptr() if d_eo is not None else None,
                                               d_ei.data_ptr() if d_ei is not None else None, ranks.data_ptr(),
                                               _host._stream_for(device))
        return ranks.cpu().numpy()

    def _batch_scores(self, sequences):
        """[n_sequences, num_items] device tensor: row r == predict(sequences[r]) (bit-identical), a tile
        of sequences per pass over the item table (csrc/slk_eval.hip); used by evaluation.sequence_mrr_score."""
        self._net.train(False)
        sequences = np.atleast_2d(sequences)
        self._check_input(sequences)
        if not isinstance(self._net, PoolNet):  # torch-side encoders: one predict() per sequence
            device = next(self._net.parameters()).device
            return torch.stack([torch.from_numpy(self.predict(row)) for row in sequences]).to(device)
        seqs = np.ascontiguousarray(sequences.astype(np.int64))
        device = self._net.tables()[0].device
        d_seqs = torch.from_numpy(seqs).to(device)
        out = torch.empty((seqs.shape[0], self._num_items), dtype=torch.float32, device=device)
        _host._engine_for(device).poolnet_scores(self._slk_tables(), d_seqs.data_ptr(), seqs.shape[0], seqs.shape[1],
                                                 out.data_ptr(), _host._stream_for(device))
        return out
