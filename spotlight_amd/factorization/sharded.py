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
