"""Embedding layers (mirror spotlight/layers.py:23-56).

They are plain parameter holders: torch owns the fp32 tables (so state_dict / pickle /
repr behave as in the reference) and the gfx950 kernels read and update them in place
through their data_ptr().  Initial values come from torch's CPU generator in the same order
as the reference, so the same seed gives the same initial tables.
"""
import torch.nn as nn


class ScaledEmbedding(nn.Embedding):
    """N(0, 1/embedding_dim) initialised embedding (layers.py:23-37)."""

    def reset_parameters(self):
        self.weight.data.normal_(0, 1.0 / self.embedding_dim)
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)


class ZeroEmbedding(nn.Embedding):
    """Zero-initialised embedding used for biases (layers.py:40-56)."""

    def reset_parameters(self):
        self.weight.data.zero_()
        if self.padding_idx is not None:
            self.weight.data[self.padding_idx].fill_(0)
