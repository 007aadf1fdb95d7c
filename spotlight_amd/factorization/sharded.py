"""Row-sharded BilinearNet training over torch.distributed.

The reference is single-device; BASELINE.json's north star asks for the item table to be
row-sharded across the GPUs of a node with RCCL all-to-all over xGMI for cross-shard row
lookups.  One process per GPU.  Rows are sharded cyclically: owner(row) = row % world, local
row = row // world, for the user AND the item tables (with their biases and optimizer state).
A rank processes the interactions of each global minibatch whose user it owns, so user rows
are always local; item rows travel by all-to-all (ids to the owners, rows back, gradient
records to the owners).  Owners sum a row's gradient contributions before ONE optimizer
update, so the semantics of the single-GPU step (factorization/implicit.py:229-243: pre-step
forward, duplicates summed) are preserved.

Everything that depends only on ids -- negatives, the sort by user, the bucketing of the
lookups by owner, the id exchange, the owners' sort by row -- is done once per CHUNK of
minibatches, so the minibatch loop issues kernels and collectives without ever waiting for
the GPU: the split sizes of every exchange of the chunk come from one count exchange (the one
host synchronisation per chunk).  Each minibatch is further cut into `slices` by user; the row
and gradient exchanges of one slice run (async_op) while another slice computes, so xGMI and
HBM work overlap.

The compute phases are the slk_shard_* entry points of include/spotlight_hip.h; this module is
the host side: buffers, split sizes and the collectives (torch.distributed = RCCL on ROCm).
"""
import torch
import torch.distributed as dist

from spotlight_amd import _native


def local_rows(num_rows, world, rank):
    """Rows of a cyclically sharded table that live on `rank`."""
    return (int(num_rows) - rank + world - 1) // world


def _bits(x):
    b = 1
    while x >> b:
        b += 1
    return b


class ShardedBilinearTrainer(object):
    """One rank's half of the row-sharded training step.

    Parameters
    ----------
    engine: _native.Engine bound to this rank's device.
    tables: the four LOCAL shards [user_emb, item_emb, user_bias, item_bias] (torch tensors on
        the engine's device; updated in place).
    optim: _native.SlkOptim over the local optimizer-state tensors (its `step` is advanced).
    num_items_global: total number of item rows (negatives are drawn over this range).
    group: process group (default: WORLD).
    stream: raw hipStream_t the kernels are enqueued on (torch's current stream).
    slices: user-slices per minibatch (exchange/compute overlap); default 4 when world > 1.
    """

    def __init__(self, engine, tables, optim, num_items_global, group=None, stream=0, slices=None, user_bias_zero=False):
        self.engine = engine
        self.tables = tables
        self.optim = optim
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_items_global = int(num_items_global)
        self.stream = stream
        self.slices = int(slices) if slices else (4 if self.world > 1 else 1)
        w = tables
        self.dim = w[0].shape[1]
        self.device = w[0].device
        # user_bias_zero: the caller has CHECKED that this rank's user biases are all zero (include/spotlight_hip.h,
        # SLK_TABLES_USER_BIAS_ZERO: bpr / hinge never move them and the user pass then does not fetch them)
        self._tables = _native.make_tables([t.data_ptr() for t in w], w[0].shape[0], w[1].shape[0], self.dim,
                                           user_bias_zero=user_bias_zero)
        self._shard = _native.make_shard(self.world, self.rank, self.num_items_global)
        self.slot_floats = self.dim + 1  # an exchange slot: a row (dim floats) + its scalar, in blocks of 64 slots
        self._bufs = {}
        self.exchange_rows = 0
        self.last_exchange_rows = 0
        # MEASURED wire bytes: what this rank handed to the collectives for OTHER ranks since train()
        # began -- ids + rows it owns + gradient rows it computed, block padding of the exchange slots included -- and the same
        # without the padding ("payload": remote lookups x (4 + 2 (D + 1) 4) bytes, the figure DESIGN.md section 7 models as
        # 0.92 KB per interaction at 8 GPUs and D = 64)
        self.exchange_bytes = 0
        self.exchange_payload_bytes = 0
        self._check_stream()

    def _check_stream(self):
        """The engine's kernels and the collectives must share ONE stream: torch.distributed's ops (RCCL) order themselves
        against torch's CURRENT stream on the device, not against the raw stream the engine was handed -- on any other stream
        the exchanges would race the kernels that fill and read their buffers, silently."""
        if self.device.type == 'cuda':
            current = torch.cuda.current_stream(self.device).cuda_stream
            if int(self.stream or 0) != int(current or 0):
                raise RuntimeError('ShardedBilinearTrainer: stream %#x is not torch\'s current stream on %s (%#x); make it current '
                                   '(torch.cuda.set_stream / with torch.cuda.stream(...)) so that the RCCL collectives are ordered '
                                   'with the engine\'s kernels' % (int(self.stream or 0), self.device, int(current or 0)))

    def bias_shadow(self, enabled=True):
        """Scope in which this rank's item biases and their Adagrad accumulator are trained interleaved (include/spotlight_hip.h:
        slk_bias_shadow_begin; the owner-side gather and the item pass index the copy).  The two tensors are stale inside it and
        rewritten on every way out; fused row-sparse Adagrad only."""
        return self.engine.bias_shadow(self._tables, self.optim, stream=self.stream, enabled=enabled)

    def max_minibatches_per_chunk(self):
        """Bound of one slk_shard_chunk_begin: (owner, unit) bins <= 2048 and 32-bit sort keys."""
        s, w = self.slices, self.world
        ub, ib = _bits(max(self.tables[0].shape[0] - 1, 1)), _bits(max(self.tables[1].shape[0] - 1, 1))
        m = 2048 // (w * s)
        m = min(m, (1 << (32 - ub)) // s, 1 << (32 - ib))
        if m < 1:
            raise ValueError('row-sharded chunk: %d slices x world %d do not fit the engine limits' % (s, w))
        return m

    def _buf(self, name, count, dtype):
        """Persistent 1-D exchange buffer, grown geometrically (a view of its first `count` elements)."""
        need = max(int(count), 1)
        b = self._bufs.get(name)
        if b is None or b.shape[0] < need:
            b = torch.empty(need + need // 4, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b[:int(count)]

    @staticmethod
    def _slots(lookups):
        """Slots a peer's segment of `lookups` lookups takes in an exchange buffer (whole blocks of
        SLK_SHARD_BLOCK = 64, include/spotlight_hip.h)."""
        return (int(lookups) + 63) // 64 * 64

    def reserve(self, batch_local, minibatches):
        """Pre-allocates the exchange buffers of chunks of `minibatches` x `batch_local` local interactions
        (uniformly spread lookups assumed, 25 % headroom), so that a timed loop allocates nothing."""
        n = int(batch_local) * int(minibatches)
        per_unit = 2 * int(batch_local) // self.slices + 1
        slack = lambda x: x + x // 4 + 1024
        padded = lambda x: (slack(x) + 64 * self.world) * self.slot_floats  # every peer's segment ends on a block boundary
        self.engine.shard_reserve(self._tables, self._shard, n, slack(2 * n))
        self._buf('send_ids', 2 * n, torch.int32)
        self._buf('recv_ids', slack(2 * n), torch.int32)
        self._buf('send_counts', self.world * minibatches * self.slices, torch.int64)
        self._buf('recv_counts', self.world * minibatches * self.slices, torch.int64)
        self._buf('grad_recv', padded(2 * int(batch_local)) + 64 * self.world * self.slices * self.slot_floats, torch.float32)
        for k in range(self.slices):
            for name in ('rows_send%d', 'rows_recv%d', 'grad_send%d'):
                self._buf(name % k, padded(per_unit), torch.float32)

    def run_chunk(self, users_local, items, mb_off, global_batches, loss='bpr', neg_in=None, neg_out=None, n_neg=None,
                  mb_pos=None):
        """A chunk of M = len(mb_off) - 1 consecutive global minibatches.  `users_local` / `items`:
        int64 device tensors with this rank's interactions of the chunk (LOCAL user rows, GLOBAL item
        ids; may be empty), minibatch m = [mb_off[m], mb_off[m + 1]); `global_batches[m]`: size of the
        GLOBAL minibatch (losses are means over it).  Negatives: `neg_in`, or one contiguous draw from
        this rank's engine RNG.  Returns a [M] tensor: this rank's share of each loss.item() (sum
        over ranks = the loss).

        loss='adaptive_hinge' (spotlight/factorization/implicit.py:266-275, spotlight/losses.py:127-166): `n_neg` draws per
        interaction (`neg_in`: [n * n_neg], interaction k's draws = entries [k * n_neg, (k + 1) * n_neg) -- its slice of its
        minibatch's ONE flat randint call), and `mb_pos`: int64 [n], every interaction's position inside its GLOBAL minibatch --
        the reference views the flat scores as [n_neg, B], so column c's candidates belong to OTHER interactions (of other
        ranks): all 1 + n_neg rows travel for a score phase, the [B, 1 + n_neg] score matrix is summed over the ranks, every
        rank runs the selection on the whole matrix, then the user pass and the gradient exchange follow as for the other
        losses (every lookup's slot travels back, zeros for the pairs the selection left out)."""
        eng, w, s_n, st = self.engine, self.world, self.slices, self.stream
        self._check_stream()
        adaptive = loss == 'adaptive_hinge'
        if adaptive and (not n_neg or mb_pos is None):
            raise ValueError('adaptive_hinge on the row-sharded path needs n_neg and mb_pos')
        NP = (int(n_neg) + 1) if adaptive else 2  # lookups per interaction
        m_n = len(mb_off) - 1
        t_n = m_n * s_n
        n = int(mb_off[-1])
        assert int(users_local.numel()) == n and int(mb_off[0]) == 0
        send_ids = self._buf('send_ids', NP * n, torch.int32)
        send_counts = self._buf('send_counts', w * t_n, torch.int64)
        if adaptive:
            assert int(mb_pos.numel()) == n and (neg_in is None or int(neg_in.numel()) == n * (NP - 1))
            eng.shard_chunk_begin_adaptive(self._tables, self._shard, users_local.data_ptr() if n else None,
                                           items.data_ptr() if n else None, n, mb_off, s_n, NP - 1,
                                           mb_pos.data_ptr() if n else None, send_ids.data_ptr(), send_counts.data_ptr(),
                                           d_neg_in=neg_in.data_ptr() if (neg_in is not None and n) else None,
                                           d_neg_out=neg_out.data_ptr() if (neg_out is not None and n) else None, stream=st)
        else:
            eng.shard_chunk_begin(self._tables, self._shard, users_local.data_ptr() if n else None,
                                  items.data_ptr() if n else None, n, mb_off, s_n, send_ids.data_ptr(),
                                  send_counts.data_ptr(),
                                  d_neg_in=neg_in.data_ptr() if (neg_in is not None and n) else None,
                                  d_neg_out=neg_out.data_ptr() if (neg_out is not None and n) else None, stream=st)
        # counts [owner][unit] -> [source][unit]; the chunk's only host synchronisation
        recv_counts = self._buf('recv_counts', w * t_n, torch.int64)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        sc_peer = [sum(sc[r * t_n:(r + 1) * t_n]) for r in range(w)]
        rc_peer = [sum(rc[r * t_n:(r + 1) * t_n]) for r in range(w)]
        recv_ids = self._buf('recv_ids', sum(rc_peer), torch.int32)
        dist.all_to_all_single(recv_ids, send_ids, rc_peer, sc_peer, group=self.group)
        eng.shard_chunk_commit(self._tables, self._shard, sc, rc, recv_ids.data_ptr(), stream=st)
        # split sizes of unit t's row / gradient exchanges, in floats: every peer's segment is whole blocks of slots
        f = self.slot_floats
        sc_unit = [[self._slots(sc[r * t_n + t]) * f for r in range(w)] for t in range(t_n)]
        rc_unit = [[self._slots(rc[r * t_n + t]) * f for r in range(w)] for t in range(t_n)]
        n_send = [sum(x) for x in sc_unit]  # floats of the unit's requester-side buffers
        n_recv = [sum(x) for x in rc_unit]  # floats of its owner-side buffers
        self.last_exchange_rows = NP * n - sc_peer[self.rank]  # lookups that crossed xGMI
        self.exchange_rows += self.last_exchange_rows
        others = [r for r in range(w) if r != self.rank]
        # ids out (4 B per remote lookup); per unit: rows this rank OWNS out to their requesters (rc_unit), gradient rows of
        # this rank's REQUESTS out to their owners (sc_unit) -- the split sizes the all_to_all_single calls below are given
        self.exchange_bytes += 4 * sum(sc_peer[r] for r in others) + 4 * sum(rc_unit[t][r] + sc_unit[t][r] for t in range(t_n) for r in others)
        self.exchange_payload_bytes += (4 * sum(sc_peer[r] for r in others) +
                                        4 * f * sum(rc[r * t_n + t] + sc[r * t_n + t] for t in range(t_n) for r in others))
        if adaptive and w > 1:
            self.exchange_bytes += 4 * NP * sum(int(g) for g in global_batches)  # the score matrices' all-reduce (sent once per ring step: a lower bound)
        loss_out = torch.zeros(m_n, dtype=torch.float32, device=self.device)
        # world 1: every lookup's owner is this rank, so an all_to_all_single would be a device-local copy of the whole buffer
        # (rcclGenericKernel: 2 x 0.43 ms per C2 minibatch, profiles/r02_o_*): the requester side reads the owner side's
        # buffers in place.  (At world > 1 the collective moves the local segment itself.)
        alias = w == 1
        for m in range(m_n):
            units = range(m * s_n, (m + 1) * s_n)
            # owners: the rows of every slice's requests; rows travel back (async)
            rows_recv, h_rows = [], []
            for k, t in enumerate(units):
                rows_send = self._buf('rows_send%d' % k, n_recv[t], torch.float32)
                eng.shard_gather(self._tables, t, rows_send.data_ptr(), stream=st)
                if alias:
                    rows_recv.append(rows_send)
                    h_rows.append(None)
                    continue
                rr = self._buf('rows_recv%d' % k, n_send[t], torch.float32)
                rows_recv.append(rr)
                h_rows.append(dist.all_to_all_single(rr, rows_send, sc_unit[t], rc_unit[t], group=self.group,
                                                     async_op=True))
            gk = None
            if adaptive:
                # score phase: this rank's rows of the minibatch's [B, 1 + n_neg] score matrix; summed over the ranks (every
                # entry has one non-zero contributor: exact); the selection, on every rank alike
                gb = int(global_batches[m])
                scores = self._buf('scores', gb * NP, torch.float32)
                scores.zero_()
                for k, t in enumerate(units):
                    if h_rows[k] is not None:
                        h_rows[k].wait()
                        h_rows[k] = None
                    eng.shard_score_pass(self._tables, t, rows_recv[k].data_ptr(), scores.data_ptr(), stream=st)
                if w > 1:
                    dist.all_reduce(scores, group=self.group)
                gk = self._buf('gk', gb * NP, torch.float32)
                eng.shard_adaptive_select(gb, NP - 1, scores.data_ptr(), gk.data_ptr(), loss_out[m:].data_ptr(),
                                          report_loss=self.rank == 0, stream=st)
            # requesters: forward / loss / backward / user update per slice; gradients travel to
            # the owners (async) while the next slice computes
            grad_recv = self._buf('grad_recv', sum(n_recv[t] for t in units), torch.float32)
            h_grad, off = [], 0
            for k, t in enumerate(units):
                if h_rows[k] is not None:
                    h_rows[k].wait()
                grad_send = grad_recv[off:off + n_recv[t]] if alias else self._buf('grad_send%d' % k, n_send[t], torch.float32)
                if adaptive:
                    eng.shard_user_pass_adaptive(self._tables, self.optim, t, gk.data_ptr(), rows_recv[k].data_ptr(),
                                                 grad_send.data_ptr(), stream=st)
                else:
                    eng.shard_user_pass(self._tables, self.optim, self._shard, t, global_batches[m], loss,
                                        rows_recv[k].data_ptr(), grad_send.data_ptr(), loss_out[m:].data_ptr(),
                                        accumulate=k > 0, stream=st)
                if not alias:
                    h_grad.append(dist.all_to_all_single(grad_recv[off:off + n_recv[t]], grad_send, rc_unit[t],
                                                         sc_unit[t], group=self.group, async_op=True))
                off += n_recv[t]
            for h in h_grad:
                h.wait()
            # owners: per unique row, sum of the minibatch's gradients, ONE optimizer update
            eng.shard_item_pass(self._tables, self.optim, m, grad_recv.data_ptr(), stream=st)
        return loss_out

    def step(self, users_local, items, global_batch, loss='bpr', neg_in=None, neg_out=None, n_neg=None, mb_pos=None):
        """One global minibatch (a chunk of one).  Returns a 1-element tensor: this rank's share of
        loss.item() (sum over ranks = the loss)."""
        return self.run_chunk(users_local, items, [0, int(users_local.numel())], [global_batch], loss=loss,
                              neg_in=neg_in, neg_out=neg_out, n_neg=n_neg, mb_pos=mb_pos)

    def train(self, users_local, items, batch_local, loss='bpr', mb_loss=None, sample_chunk=8, n_neg=None):
        """Minibatch loop over this rank's interactions: global minibatch k consists of every
        rank's slice [k*batch_local, (k+1)*batch_local) (all ranks must hold the same number of
        interactions).  Negatives are drawn from this rank's engine RNG over the global item range,
        one contiguous draw per chunk of `sample_chunk` minibatches (exactly what per-minibatch
        draws would produce: sampling.py:34 draws are independent per output).
        Returns the per-minibatch loss shares (sum over ranks = loss.item())."""
        n = int(users_local.numel())
        n_mb = (n + batch_local - 1) // batch_local
        if mb_loss is None:
            mb_loss = torch.zeros(n_mb, dtype=torch.float32, device=self.device)
        self.exchange_rows = 0
        self.exchange_bytes = self.exchange_payload_bytes = 0
        per_chunk = max(1, min(int(sample_chunk), self.max_minibatches_per_chunk()))
        for k0 in range(0, n_mb, per_chunk):
            k1 = min(k0 + per_chunk, n_mb)
            lo, hi = k0 * batch_local, min(k1 * batch_local, n)
            off = [min(k * batch_local, n) - lo for k in range(k0, k1 + 1)]
            gbs = [(off[i + 1] - off[i]) * self.world for i in range(k1 - k0)]
            mb_pos = None
            if loss == 'adaptive_hinge':
                # equal shares: the j-th interaction this rank holds of a minibatch is the global minibatch's (j * world + rank)-th
                j = torch.arange(lo, hi, device=self.device, dtype=torch.int64) % batch_local
                mb_pos = j * self.world + self.rank
            mb_loss[k0:k1].copy_(self.run_chunk(users_local[lo:hi], items[lo:hi], off, gbs, loss=loss, n_neg=n_neg, mb_pos=mb_pos))
        return mb_loss


# ---------------------------------------------------------------------------------------------
# The drop-in model on top of the trainer
# ---------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

from spotlight_amd.factorization import implicit as _host  # noqa: E402
from spotlight_amd.factorization._components import _predict_process_ids  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.factorization.representations import BilinearNet  # noqa: E402

_FULL_INIT_LIMIT_BYTES = 8 << 30


class ShardedImplicitFactorizationModel(ImplicitFactorizationModel):
    """ImplicitFactorizationModel whose four tables are row-sharded over the ranks of a
    torch.distributed process group (one process per GPU; `torchrun`).

    Same constructor, `fit(interactions)` and `predict(user_ids, item_ids=None)` as the
    single-device model (spotlight/factorization/implicit.py:76-311); every rank makes the same
    calls with the same arguments (SPMD) and gets the same return values.  Given the same
    `random_state` seed on every rank, a run consumes the RandomState exactly like the
    single-device model: same model seed draw, same shuffles, and the SAME negatives (every rank
    draws the epoch's whole negative stream on its GPU and keeps the entries of its own
    interactions), so the trained tables match a single-device run to summation-order noise.
    Initial values also match when the full tables fit the host (< 8 GB): they are drawn from
    torch's CPU generator in the reference's order and this rank keeps rows `rank::world`.

    All four losses (adaptive hinge: a score phase and a selection over the whole minibatch in front of the user pass,
    ShardedBilinearTrainer.run_chunk).  Restriction of the exchange path: plain (non-bloom) tables.
    """

    # evaluation.mrr_score's one-device fast path scores against whole tables; this model's are local shards
    # indexed by local rows, so ranking goes through predict() (rows assembled from their owners)
    _batch_scores = None
    _fused_ranks = None

    def __init__(self, *args, **kwargs):
        self._group = kwargs.pop('group', None)
        super(ShardedImplicitFactorizationModel, self).__init__(*args, **kwargs)
        if self._representation is not None:
            raise NotImplementedError('custom representations are not supported by the row-sharded path')
        self._trainer = None

    def __getstate__(self):
        state = super(ShardedImplicitFactorizationModel, self).__getstate__()
        state['_trainer'] = None
        state['_group'] = None
        return state

    @property
    def _world(self):
        return dist.get_world_size(self._group)

    @property
    def _rank(self):
        return dist.get_rank(self._group)

    def _initialize(self, interactions):
        self._num_users, self._num_items = interactions.num_users, interactions.num_items
        world, rank, D = self._world, self._rank, self._embedding_dim
        U, I = self._num_users, self._num_items
        full = None
        if (U + I) * D * 4 <= _FULL_INIT_LIMIT_BYTES:
            full = BilinearNet(U, I, D, sparse=self._sparse)  # the reference's draws, in its order
        net = BilinearNet(local_rows(U, world, rank), local_rows(I, world, rank), D, sparse=self._sparse)
        if full is not None:
            with torch.no_grad():
                for loc, whole in zip(net.tables(), full.tables()):
                    loc.copy_(whole[rank::world])
        self._net = net.to(_host._model_device())
        if self._optimizer_func is None:
            self._optimizer = torch.optim.Adam(self._net.parameters(), weight_decay=self._l2,
                                               lr=self._learning_rate)
        else:
            self._optimizer = self._optimizer_func(self._net.parameters())
        self._loss_func = self._loss
        self._binding = None
        self._trainer = None

    def fit(self, interactions, verbose=False):
        user_ids, item_ids = interactions.user_ids, interactions.item_ids
        if not self._initialized:
            self._initialize(interactions)
        self._check_input(user_ids, item_ids)

        binding = self._bind()
        tables = self._net.tables()
        device = tables[0].device
        engine = _host._engine_for(device)
        stream = _host._stream_for(device)
        world, rank, B = self._world, self._rank, self._batch_size
        n = len(user_ids)
        n_mb = (n + B - 1) // B

        d_users0 = _host.ids_to_device(user_ids, device)
        d_items0 = _host.ids_to_device(item_ids, device)
        d_users, d_items = torch.empty_like(d_users0), torch.empty_like(d_items0)
        d_perm = torch.empty(n, dtype=torch.int64, device=device)
        # large local item shards train with {bias, Adagrad accumulator} interleaved for the duration of fit(), as the one-GPU
        # model does (factorization/implicit.py: _BIAS_SHADOW_MIN_ITEMS); the scope's end -- also on an exception -- writes both back
        shadow = engine.bias_shadow(_native.make_tables([t.data_ptr() for t in tables], tables[0].shape[0], tables[1].shape[0],
                                                        tables[0].shape[1]), binding.as_struct(), stream=stream,
                                    enabled=binding.kind == 'adagrad' and tables[1].shape[0] >= _host._BIAS_SHADOW_MIN_ITEMS)
        with shadow:
            self._fit_epochs(binding, engine, tables, device, stream, d_users0, d_items0, d_users, d_items, d_perm, n, n_mb, verbose)

    def _fit_epochs(self, binding, engine, tables, device, stream, d_users0, d_items0, d_users, d_items, d_perm, n, n_mb, verbose):
        world, rank, B = self._world, self._rank, self._batch_size
        for epoch_num in range(self._n_iter):
            # every rank computes the same numpy-exact permutation and negatives on its own GPU
            engine.rng_set_state(self._random_state.get_state())
            _host.device_epoch_shuffle(engine, self._random_state, n, d_perm,
                                       [(d_users0, d_users, 1), (d_items0, d_items, 1)], stream)
            # the epoch's negatives: one randint per minibatch == one contiguous draw over the epoch (adaptive hinge: B * n draws
            # per minibatch, interaction j of it scored against the flat entries [j * n, (j + 1) * n): implicit.py:266-275)
            adaptive = self._loss == 'adaptive_hinge'
            nn = self._num_negative_samples if adaptive else 1
            negs = torch.empty(n * nn, dtype=torch.int64, device=device)
            engine.sample_items(self._num_items, n * nn, negs.data_ptr(), stream=stream)
            self._random_state.set_state(engine.rng_get_state())
            # this rank's interactions, minibatch membership unchanged
            idx = torch.nonzero(d_users % world == rank).squeeze(1)
            bounds = torch.searchsorted(idx, torch.arange(0, n_mb + 1, device=device) * B).tolist()
            ul = (d_users[idx] // world).contiguous()
            il = d_items[idx].contiguous()
            ng = negs.view(n, nn)[idx].contiguous().view(-1)
            pos = (idx % B).contiguous() if adaptive else None  # position inside the global minibatch
            ostruct = binding.as_struct()
            trainer = ShardedBilinearTrainer(engine, tables, ostruct, self._num_items, group=self._group,
                                             stream=stream) if self._trainer is None else self._trainer
            trainer.optim = ostruct
            # (this rank's user biases checked on the device each epoch: all zero under bpr / hinge for a model initialised as the
            # reference initialises it -- the user pass then does not fetch them, implicit.py::_fit)
            hint = (self._loss in ('bpr', 'hinge') and binding.kind in ('adagrad', 'sgd') and not bool(tables[2].any()))
            trainer._tables.flags = _native.TABLES_USER_BIAS_ZERO if hint else 0
            trainer.stream = stream  # (a later fit() under another current stream: _check_stream validates the one in use)
            self._trainer = trainer
            mb_loss = torch.zeros(n_mb, dtype=torch.float32, device=device)
            per_chunk = min(8, trainer.max_minibatches_per_chunk())
            for k0 in range(0, n_mb, per_chunk):
                k1 = min(k0 + per_chunk, n_mb)
                a, b = bounds[k0], bounds[k1]
                off = [bounds[k] - a for k in range(k0, k1 + 1)]
                gbs = [min(B, n - k * B) for k in range(k0, k1)]
                mb_loss[k0:k1].copy_(trainer.run_chunk(ul[a:b], il[a:b], off, gbs, loss=self._loss, neg_in=ng[a * nn:b * nn],
                                                       n_neg=nn if adaptive else None, mb_pos=pos[a:b] if adaptive else None))
            dist.all_reduce(mb_loss, group=self._group)
            binding.store_steps(ostruct.step)
            epoch_loss = float(mb_loss.double().mean().item())
            if verbose and rank == 0:
                print('Epoch {}: loss {}'.format(epoch_num, epoch_loss))
            if np.isnan(epoch_loss) or epoch_loss == 0.0:
                raise ValueError('Degenerate epoch loss: {}'.format(epoch_loss))

    def _fetch_rows(self, t_emb, t_bias, ids, device):
        """[len(ids), D + 1]: embedding row and bias of every id, assembled from the owners."""
        world, rank = self._world, self._rank
        w = self._net.tables()
        d_ids = torch.from_numpy(np.ascontiguousarray(ids)).to(device)
        out = torch.zeros(d_ids.numel(), w[t_emb].shape[1] + 1, dtype=torch.float32, device=device)
        mine = torch.nonzero(d_ids % world == rank).squeeze(1)
        loc = d_ids[mine] // world
        out[mine, :-1] = w[t_emb].detach()[loc]
        out[mine, -1] = w[t_bias].detach()[loc, 0]
        dist.all_reduce(out, group=self._group)  # every row has exactly one owner
        return out

    def predict(self, user_ids, item_ids=None):
        self._check_input(user_ids, item_ids, allow_items_none=True)
        self._net.train(False)
        users, items, n = _predict_process_ids(user_ids, item_ids, self._num_items)
        if items is None:
            items = np.arange(n, dtype=np.int64)
        device = self._net.tables()[0].device
        engine = _host._engine_for(device)
        ru = self._fetch_rows(0, 2, users.reshape(-1), device)
        ri = self._fetch_rows(1, 3, items.reshape(-1), device)
        D = ru.shape[1] - 1
        gathered = [ru[:, :D].contiguous(), ri[:, :D].contiguous(), ru[:, D].contiguous(), ri[:, D].contiguous()]
        tb = _native.make_tables([t.data_ptr() for t in gathered], ru.shape[0], ri.shape[0], D)
        d_u = torch.arange(ru.shape[0], dtype=torch.int64, device=device)
        out = torch.empty(n, dtype=torch.float32, device=device)
        engine.bilinear_predict(tb, d_u.data_ptr(), ru.shape[0], None, n, out.data_ptr(), _host._stream_for(device))
        return out.cpu().numpy().flatten()
