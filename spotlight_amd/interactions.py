"""Interaction containers (drop-in for spotlight/interactions.py:38-312).

Pure host numpy, same constructor signatures, attributes, validation messages and
`to_sequence` output as the reference.  `to_sequence(..., device='cuda')` builds the same matrix
with the HIP kernels of csrc/slk_seqprep.hip instead of the host double loop (no CPU fallback on
that route: it raises if the HIP library or device is missing).
"""
import numpy as np
import scipy.sparse as sp


class Interactions(object):
    """(user, item[, rating, timestamp, weight]) tuples.  interactions.py:95-168."""

    def __init__(self, user_ids, item_ids, ratings=None, timestamps=None, weights=None,
                 num_users=None, num_items=None):
        self.num_users = num_users or int(user_ids.max() + 1)
        self.num_items = num_items or int(item_ids.max() + 1)
        self.user_ids = user_ids
        self.item_ids = item_ids
        self.ratings = ratings
        self.timestamps = timestamps
        self.weights = weights
        self._check()

    def __repr__(self):
        return ('<Interactions dataset ({num_users} users x {num_items} items '
                'x {num_interactions} interactions)>'
                .format(num_users=self.num_users, num_items=self.num_items,
                        num_interactions=len(self)))

    def __len__(self):
        return len(self.user_ids)

    def _check(self):
        if self.user_ids.max() >= self.num_users:
            raise ValueError('Maximum user id greater than declared number of users.')
        if self.item_ids.max() >= self.num_items:
            raise ValueError('Maximum item id greater than declared number of items.')
        n = len(self.user_ids)
        for name, value in (('item IDs', self.item_ids), ('ratings', self.ratings),
                            ('timestamps', self.timestamps), ('weights', self.weights)):
            if value is not None and len(value) != n:
                raise ValueError('Invalid {} dimensions: length must be equal to number of '
                                 'interactions'.format(name))

    def tocoo(self):
        data = self.ratings if self.ratings is not None else np.ones(len(self))
        return sp.coo_matrix((data, (self.user_ids, self.item_ids)),
                             shape=(self.num_users, self.num_items))

    def tocsr(self):
        return self.tocoo().tocsr()

    def to_sequence(self, max_sequence_length=10, min_sequence_length=None, step_size=None, device=None):
        """Left-zero-padded (num_sequences x max_sequence_length) windows over each user's
        time-ordered items, newest window first (interactions.py:170-266).  Windows end at
        positions len, len-step, len-2*step, ... of the user's history.

        device=None: on the host, as the reference.  device='cuda' (or a torch.device): sorted and
        cut on the GPU (slk_to_sequence_plan / slk_to_sequence_fill), identical output."""
        if self.timestamps is None:
            raise ValueError('Cannot convert to sequences, timestamps not available.')
        if 0 in self.item_ids:
            raise ValueError('0 is used as an item id, conflicting with the sequence padding value.')
        if step_size is None:
            step_size = max_sequence_length
        if device is not None:
            return self._to_sequence_device(max_sequence_length, min_sequence_length, step_size, device)
        order = np.lexsort((self.timestamps, self.user_ids))
        users = self.user_ids[order]
        items = self.item_ids[order]
        uniq, starts, counts = np.unique(users, return_index=True, return_counts=True)
        per_user = -(-counts // step_size)  # ceil
        total = int(per_user.sum())
        sequences = np.zeros((total, max_sequence_length), dtype=np.int32)
        sequence_users = np.empty(total, dtype=np.int32)
        row = 0
        for uid, start, count in zip(uniq, starts, counts):
            hist = items[start:start + count]
            for end in range(count, 0, -step_size):
                window = hist[max(end - max_sequence_length, 0):end]
                sequences[row, max_sequence_length - len(window):] = window
                sequence_users[row] = uid
                row += 1
        if min_sequence_length is not None:
            keep = sequences[:, -min_sequence_length] != 0
            sequences, sequence_users = sequences[keep], sequence_users[keep]
        return SequenceInteractions(sequences, user_ids=sequence_users, num_items=self.num_items)


    def _to_sequence_device(self, max_sequence_length, min_sequence_length, step_size, device):
        import torch

        from spotlight_amd.factorization import implicit as _host
        requested = torch.device(device)
        if requested.type != 'cuda':
            raise ValueError("to_sequence(device=...) needs a HIP device ('cuda'); use device=None for the host route")
        device = requested if requested.index is not None else _host._model_device()
        L = int(max_sequence_length)
        # the reference's filter is `sequences[:, -min_sequence_length] != 0`: with the item ids all
        # non-zero that keeps the windows holding at least min_length items
        if min_sequence_length is None:
            min_length = 1
        else:
            m = int(min_sequence_length)
            column = L - m if m > 0 else -m
            if not 0 <= column < L:
                raise IndexError('index %d is out of bounds for axis 1 with size %d' % (-m, L))
            min_length = L - column
        ts = np.ascontiguousarray(self.timestamps)
        if ts.dtype.kind in 'iub':
            ts, ts_kind = ts.astype(np.int64, copy=False), 0
        elif ts.dtype.kind == 'f':
            ts, ts_kind = ts.astype(np.float64, copy=False), 1
        else:
            raise TypeError('to_sequence(device=...): timestamps must be integers or floats, got %s' % ts.dtype)
        engine = _host._engine_for(device)
        stream = _host._stream_for(device)
        d_users = _host.ids_to_device(self.user_ids, device)
        d_items = _host.ids_to_device(self.item_ids, device)
        d_ts = torch.from_numpy(ts).to(device)
        rows = engine.to_sequence_plan(d_users.data_ptr(), d_items.data_ptr(), d_ts.data_ptr(), ts_kind, len(self),
                                       self.num_users, L, int(step_size), min_length, stream)
        d_seq = torch.empty((rows, L), dtype=torch.int32, device=device)
        d_seq_users = torch.empty((rows,), dtype=torch.int32, device=device)
        engine.to_sequence_fill(d_seq.data_ptr(), d_seq_users.data_ptr(), stream)
        return SequenceInteractions(d_seq.cpu().numpy(), user_ids=d_seq_users.cpu().numpy(), num_items=self.num_items)


class SequenceInteractions(object):
    """Sequence matrix container (interactions.py:269-312)."""

    def __init__(self, sequences, user_ids=None, num_items=None):
        self.sequences = sequences
        self.user_ids = user_ids
        self.max_sequence_length = sequences.shape[1]
        self.num_items = sequences.max() + 1 if num_items is None else num_items

    def __repr__(self):
        n, length = self.sequences.shape
        return ('<Sequence interactions dataset ({num_sequences} sequences x {sequence_length} '
                'sequence length)>'.format(num_sequences=n, sequence_length=length))
