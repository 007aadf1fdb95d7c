"""Embedding layers (mirror spotlight/layers.py:23-56).

They are parameter holders: torch owns the fp32 tables (so state_dict / pickle / repr behave
as in the reference) and the fused training kernels read and update them in place through
their data_ptr().  Initial values come from torch's CPU generator in the same order as the
reference, so the same seed gives the same initial tables.  Calling a layer (`layer(ids)`, what
the LSTM / CNN / mixture encoders do) goes through spotlight_amd.embedding.lookup: the gather,
the bloom hashed-row sum and their autograd backward are gfx950 kernels (csrc/slk_embed.hip).
"""
import torch.nn as nn

from spotlight_amd.embedding import lookup


class ScaledEmbedding(nn.Embedding):
    """N(0, 1/embedding_dim) initialised embedding (layers.py:23-37)."""

    def reset_parameters(self):
        self.weight.data.normal_(0, 1.0 / self.embedding_dim)
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)

    def forward(self, indices):
        return lookup(self.weight, indices, padding_idx=self.padding_idx, sparse=self.sparse)


class ZeroEmbedding(nn.Embedding):
    """Zero-initialised embedding used for biases (layers.py:40-56)."""

    def reset_parameters(self):
        self.weight.data.zero_()
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)

    def forward(self, indices):
        return lookup(self.weight, indices, padding_idx=self.padding_idx, sparse=self.sparse)


class ScaledEmbeddingBag(nn.EmbeddingBag):
    """N(0, 1/embedding_dim) initialised EmbeddingBag (layers.py:59-71).  Only the reference's
    BloomEmbedding(bag=True) uses it, a mode this package rejects (see BloomEmbedding); the class is kept
    for code that names it.  It is a stock torch module: no kernel of this package is involved."""

    def reset_parameters(self):
        self.weight.data.normal_(0, 1.0 / self.embedding_dim)


SEEDS = [
    179424941, 179425457, 179425907, 179426369,
    179424977, 179425517, 179425943, 179426407,
    179424989, 179425529, 179425993, 179426447,
    179425003, 179425537, 179426003, 179426453,
    179425019, 179425559, 179426029, 179426491,
    179425027, 179425579, 179426081, 179426549
]


class BloomEmbedding(nn.Module):
    """Hashed (bloom) embedding layer -- mirrors spotlight/layers.py:74-244.

    The embedding of index x is the sum of `num_hash_functions` rows of a compressed table
    with int(compression_ratio * num_embeddings) rows, selected by MurmurHash3 with the
    reference's seeds; index `padding_idx` maps to row 0, which is zero and never trained.
    The hashing, the gather-and-sum, the backward into the hashed rows and the optimizer
    update run inside the fused gfx950 kernels (csrc/slk_kernels.h: slk_emb_vec, hashed-row
    owner passes); hashes are recomputed in-kernel instead of being cached per index.

    `bag=True` is not supported: the reference builds its EmbeddingBag offsets with stride 1
    instead of num_hash_functions (layers.py:219-222), so that mode does not compute the
    documented sum -- there is nothing well-defined to reproduce.
    """

    def __init__(self, num_embeddings, embedding_dim, compression_ratio=0.2, num_hash_functions=4,
                 bag=False, padding_idx=0):
        super(BloomEmbedding, self).__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.compression_ratio = compression_ratio
        self.compressed_num_embeddings = int(compression_ratio * num_embeddings)
        self.num_hash_functions = num_hash_functions
        self.padding_idx = padding_idx
        self._bag = bag
        if num_hash_functions > len(SEEDS):
            raise ValueError('Can use at most {} hash functions ({} requested)'
                             .format(len(SEEDS), num_hash_functions))
        if num_hash_functions > 8:
            raise NotImplementedError('the gfx950 kernels support at most 8 hash functions')
        if bag:
            raise NotImplementedError('BloomEmbedding(bag=True) is not supported (see class docstring)')
        self._masks = SEEDS[:self.num_hash_functions]
        self.embeddings = ScaledEmbedding(self.compressed_num_embeddings, self.embedding_dim,
                                          padding_idx=self.padding_idx)

    def __repr__(self):
        return ('<BloomEmbedding (compression_ratio: {}): {}>'
                .format(self.compression_ratio, repr(self.embeddings)))

    @property
    def weight(self):
        """The compressed table (what the kernels read and update in place)."""
        return self.embeddings.weight

    def forward(self, indices):
        """[batch] -> [batch, 1, dim]; [batch, seq] -> [batch, seq, dim]: sums of the hashed rows
        (layers.py:200-244), hashed in-kernel."""
        if indices.dim() == 2:
            batch_size, seq_size = indices.size()
        else:
            batch_size, seq_size = indices.size(0), 1
        out = lookup(self.embeddings.weight, indices.reshape(batch_size * seq_size), bloom=self.descriptor(),
                     padding_idx=self.padding_idx, sparse=self.embeddings.sparse)
        return out.view(batch_size, seq_size, -1)

    def descriptor(self):
        """include/spotlight_hip.h: slk_bloom for this layer."""
        from spotlight_amd import _native
        return _native.make_bloom(self.compressed_num_embeddings, self.num_hash_functions,
                                  padding_idx=self.padding_idx, skip_row=self.padding_idx, seeds=self._masks)
