"""Row-sharded BilinearNet training over torch.distributed.

The reference is single-device; BASELINE.json's north star asks for the item table to be
row-sharded across the GPUs of a node with RCCL all-to-all over xGMI for cross-shard row
lookups.  One process per GPU.  Rows are sharded cyclically: owner(row) = row % world, local
row = row // world, for the user AND the item tables (with their biases and optimizer state).
A rank processes the interactions of each global minibatch whose user it owns, so user rows
are always local; item rows travel in three all-to-all phases per minibatch (ids to the
owners, rows back, gradient records to the owners).  Owners sum a row's gradient
contributions before ONE optimizer update, so the semantics of the single-GPU step
(factorization/implicit.py:229-243: pre-step forward, duplicates summed) are preserved.

The compute phases are the slk_shard_* entry points of include/spotlight_hip.h; this module is
the host side: buffers, split sizes and the collectives (torch.distributed = RCCL on ROCm).
"""
import torch
import torch.distributed as dist

from spotlight_amd import _native


def local_rows(num_rows, world, rank):
    """Rows of a cyclically sharded table that live on `rank`."""
    return (int(num_rows) - rank + world - 1) // world


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
    """

    def __init__(self, engine, tables, optim, num_items_global, group=None, stream=0):
        self.engine = engine
        self.tables = tables
        self.optim = optim
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_items_global = int(num_items_global)
        self.stream = stream
        w = tables
        self.dim = w[0].shape[1]
        self.device = w[0].device
        self._tables = _native.make_tables([t.data_ptr() for t in w], w[0].shape[0], w[1].shape[0], self.dim)
        self.rsv = engine.shard_row_floats(self.dim)
        self._bufs = {}
        self.last_exchange_rows = 0

    def _buf(self, name, rows, cols, dtype):
        """Persistent exchange buffer, grown geometrically (views of the first `rows` rows)."""
        need = max(int(rows), 1)
        b = self._bufs.get(name)
        if b is None or b.shape[0] < need:
            cap = need + need // 4
            shape = (cap, cols) if cols else (cap,)
            b = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b[:int(rows)]

    def step(self, users_local, items, global_batch, loss='bpr', neg_in=None, neg_out=None):
        """One global minibatch.  `users_local` / `items`: int64 device tensors holding this
        rank's interactions (LOCAL user rows, GLOBAL item ids); may be empty.  Returns a
        1-element tensor: this rank's share of loss.item() (sum over ranks = the loss)."""
        eng, w = self.engine, self.world
        n = int(users_local.numel())
        sh = _native.make_shard(w, self.rank, self.num_items_global, global_batch)
        send_ids = self._buf('send_ids', 2 * n, 0, torch.int64)
        send_counts = self._buf('send_counts', w, 0, torch.int64)
        eng.shard_begin(self._tables, sh, users_local.data_ptr() if n else None, items.data_ptr() if n else None,
                        n, send_ids.data_ptr(), send_counts.data_ptr(),
                        d_neg_in=neg_in.data_ptr() if (neg_in is not None and n) else None,
                        d_neg_out=neg_out.data_ptr() if (neg_out is not None and n) else None,
                        stream=self.stream)
        # a2a #1: how many lookups each owner receives, then the owner-local row ids
        recv_counts = self._buf('recv_counts', w, 0, torch.int64)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()  # one host sync per minibatch
        n_recv = sum(rc)
        recv_ids = self._buf('recv_ids', n_recv, 0, torch.int64)
        dist.all_to_all_single(recv_ids, send_ids, rc, sc, group=self.group)
        # a2a #2: owners gather the requested rows (+ bias) and send them back
        rows_send = self._buf('rows_send', n_recv, self.rsv, torch.float32)
        eng.shard_gather(self._tables, recv_ids.data_ptr() if n_recv else None, n_recv,
                         rows_send.data_ptr() if n_recv else None, stream=self.stream)
        rows_recv = self._buf('rows_recv', 2 * n, self.rsv, torch.float32)
        dist.all_to_all_single(rows_recv, rows_send, sc, rc, group=self.group)
        # forward / loss / backward / user update on the requester
        grad_send = self._buf('grad_send', 2 * n, self.rsv, torch.float32)
        loss_out = self._buf('loss_out', 1, 0, torch.float32)
        eng.shard_user_pass(self._tables, self.optim, sh, n, loss, rows_recv.data_ptr() if n else None,
                            grad_send.data_ptr() if n else None, loss_out.data_ptr(), stream=self.stream)
        # a2a #3: gradient records to the owners, which sum per row and update once
        grad_recv = self._buf('grad_recv', n_recv, self.rsv, torch.float32)
        dist.all_to_all_single(grad_recv, grad_send, rc, sc, group=self.group)
        eng.shard_item_pass(self._tables, self.optim, recv_ids.data_ptr() if n_recv else None,
                            grad_recv.data_ptr() if n_recv else None, n_recv, stream=self.stream)
        self.last_exchange_rows = 2 * n - sc[self.rank]  # lookups that crossed xGMI
        return loss_out.clone()

    def train(self, users_local, items, batch_local, loss='bpr', mb_loss=None, sample_chunk=8):
        """Minibatch loop over this rank's interactions: global minibatch k consists of every
        rank's slice [k*batch_local, (k+1)*batch_local) (all ranks must hold the same number of
        interactions).  Negatives are drawn from this rank's engine RNG over the global item range,
        `sample_chunk` minibatches per draw (one contiguous randint stream per rank, exactly as a
        per-minibatch draw would produce: sampling.py:34 draws are independent per output).
        Returns the per-minibatch loss shares (sum over ranks = loss.item())."""
        n = int(users_local.numel())
        n_mb = (n + batch_local - 1) // batch_local
        if mb_loss is None:
            mb_loss = torch.zeros(n_mb, dtype=torch.float32, device=self.device)
        self.exchange_rows = 0
        negs = None
        for k in range(n_mb):
            lo, hi = k * batch_local, min((k + 1) * batch_local, n)
            if k % sample_chunk == 0:
                c_hi = min((k + sample_chunk) * batch_local, n)
                negs = self._buf('negs', c_hi - lo, 0, torch.int64)
                self.engine.sample_items(self.num_items_global, c_hi - lo, negs.data_ptr(), stream=self.stream)
                c_lo = lo
            part = self.step(users_local[lo:hi], items[lo:hi], (hi - lo) * self.world, loss=loss,
                             neg_in=negs[lo - c_lo:hi - c_lo])
            mb_loss[k:k + 1].copy_(part)
            self.exchange_rows += self.last_exchange_rows
        return mb_loss


# ---------------------------------------------------------------------------------------------
# The drop-in model on top of the trainer
# ---------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

from spotlight_amd.factorization import implicit as _host  # noqa: E402
from spotlight_amd.factorization._components import _predict_process_ids  # noqa: E402
from spotlight_amd.factorization.implicit import ImplicitFactorizationModel  # noqa: E402
from spotlight_amd.factorization.representations import BilinearNet  # noqa: E402
from spotlight_amd.torch_utils import shuffle  # noqa: E402

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

    Restrictions of the exchange path: pointwise / bpr / hinge losses, plain (non-bloom) tables.
    """

    def __init__(self, *args, **kwargs):
        self._group = kwargs.pop('group', None)
        super(ShardedImplicitFactorizationModel, self).__init__(*args, **kwargs)
        if self._loss == 'adaptive_hinge':
            raise NotImplementedError('adaptive_hinge is not supported by the row-sharded path yet')
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
        user_ids = interactions.user_ids.astype(np.int64)
        item_ids = interactions.item_ids.astype(np.int64)
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

        for epoch_num in range(self._n_iter):
            users, items = shuffle(user_ids, item_ids, random_state=self._random_state)
            d_users = torch.from_numpy(users).to(device)
            d_items = torch.from_numpy(items).to(device)
            # the epoch's negatives: one randint per minibatch == one contiguous draw over the epoch
            negs = torch.empty(n, dtype=torch.int64, device=device)
            engine.rng_set_state(self._random_state.get_state())
            engine.sample_items(self._num_items, n, negs.data_ptr(), stream=stream)
            self._random_state.set_state(engine.rng_get_state())
            # this rank's interactions, minibatch membership unchanged
            idx = torch.nonzero(d_users % world == rank).squeeze(1)
            bounds = torch.searchsorted(idx, torch.arange(0, n_mb + 1, device=device) * B).tolist()
            ul = (d_users[idx] // world).contiguous()
            il = d_items[idx].contiguous()
            ng = negs[idx].contiguous()
            ostruct = binding.as_struct()
            trainer = ShardedBilinearTrainer(engine, tables, ostruct, self._num_items, group=self._group,
                                             stream=stream) if self._trainer is None else self._trainer
            trainer.optim = ostruct
            self._trainer = trainer
            mb_loss = torch.zeros(n_mb, dtype=torch.float32, device=device)
            for k in range(n_mb):
                a, b = bounds[k], bounds[k + 1]
                share = trainer.step(ul[a:b], il[a:b], min(B, n - k * B), loss=self._loss, neg_in=ng[a:b])
                mb_loss[k:k + 1].copy_(share)
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
