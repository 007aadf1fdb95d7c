"""Sequence representations (mirror spotlight/sequence/representations.py:26-596).

PoolNet is a parameter holder, like factorization.representations.BilinearNet: two tables created
in the reference's order (item_embeddings: ScaledEmbedding with padding_idx=0, item_biases:
ZeroEmbedding with padding_idx=0), so state_dict keys and the torch-generator initialisation
match.  Its user_representation / forward are provided for API parity (prediction-time use, no
autograd): training's forward, backward and update are the fused kernels of csrc/slk_seq.hip.

LSTMNet, CNNNet and MixtureLSTMNet keep their recurrent / convolutional body on stock
PyTorch-ROCm (MIOpen) and train through autograd; what this package contributes to them is the
embedding front-end: `self.item_embeddings(ids)` / `self.item_biases(ids)` and their backward are
the kernels of csrc/slk_embed.hip (spotlight_amd/embedding.py), including in-kernel hashing for
BloomEmbedding layers.  Parameters are created in the reference's order under the reference's
names, so a given seed initialises them identically and state_dicts interchange.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from spotlight_amd.layers import BloomEmbedding, ScaledEmbedding, ZeroEmbedding

PADDING_IDX = 0


class PoolNet(nn.Module):

    def __init__(self, num_items, embedding_dim=32, item_embedding_layer=None, sparse=False):
        super(PoolNet, self).__init__()
        self.embedding_dim = embedding_dim
        if item_embedding_layer is not None:
            self.item_embeddings = item_embedding_layer
        else:
            self.item_embeddings = ScaledEmbedding(num_items, embedding_dim, padding_idx=PADDING_IDX,
                                                   sparse=sparse)
        self.item_biases = ZeroEmbedding(num_items, 1, sparse=sparse, padding_idx=PADDING_IDX)

    def tables(self):
        """[item_embeddings.weight, item_biases.weight] (ABI slots 1 and 3 of slk_tables)."""
        return [self.item_embeddings.weight, self.item_biases.weight]

    def _embed(self, ids):
        """Embedding vectors of `ids` (any shape) -> [..., D]: the layer's own lookup, i.e. the gfx950 gather / in-kernel
        hashed-row sum of csrc/slk_embed.hip (spotlight_amd/embedding.py) for plain and BloomEmbedding layers alike."""
        from spotlight_amd.embedding import lookup
        layer = self.item_embeddings
        if isinstance(layer, BloomEmbedding):
            return lookup(layer.weight, ids, bloom=layer.descriptor(), padding_idx=layer.padding_idx)
        return lookup(layer.weight, ids, padding_idx=getattr(layer, 'padding_idx', None))

    def user_representation(self, item_sequences):
        """(all_representations [B, D, L], final_representation [B, D]) as in the reference
        (:76-114): running sums of the item embeddings divided by (per-dimension non-zero
        count + 1).  Plain torch ops on the tables' device; not differentiable here."""
        with torch.no_grad():
            emb = self._embed(item_sequences).permute(0, 2, 1)  # [B, D, L]
            emb = torch.nn.functional.pad(emb, (1, 0))
            sums = torch.cumsum(emb, 2)
            counts = torch.cumsum((emb != 0.0).float(), 2)
            rep = sums / (counts + 1)
        return rep[:, :, :-1], rep[:, :, -1]

    def forward(self, user_representations, targets):
        """predictions[b, t] = bias[target] + <representation[b, :, t], E[target]> (:116-144)."""
        with torch.no_grad():
            w = self._embed(targets)                          # [B, L, D] or [B, 1, D]
            b = self.item_biases.weight[targets].squeeze(-1)  # [B, L]
            if user_representations.dim() == 2:
                user_representations = user_representations.unsqueeze(2)
            dot = (user_representations.permute(0, 2, 1) * w).sum(2)
            return (b + dot).squeeze()


def _as_tuple(value, length):
    """An int repeated `length` times, or the tuple itself (kernel widths / dilations per layer)."""
    if isinstance(value, (tuple, list)):
        return tuple(value)
    return (value,) * length


class _EncoderNet(nn.Module):
    """What the three torch-side encoders share: the two item tables (created first, in the reference's
    order) and the time-major [batch, 1 + seq_len, dim] embedding of a sequence behind one zero step."""

    def _make_item_tables(self, num_items, embedding_dim, item_embedding_layer, sparse):
        self.embedding_dim = embedding_dim
        if item_embedding_layer is None:
            item_embedding_layer = ScaledEmbedding(num_items, embedding_dim, padding_idx=PADDING_IDX,
                                                   sparse=sparse)
        self.item_embeddings = item_embedding_layer
        self.item_biases = ZeroEmbedding(num_items, 1, sparse=sparse, padding_idx=PADDING_IDX)

    def _shifted_embeddings(self, item_sequences):
        """[B, L] ids -> [B, L + 1, D]: step 0 is the all-zero "nothing seen yet" input."""
        return F.pad(self.item_embeddings(item_sequences), (0, 0, 1, 0))

    def _target_tables(self, targets):
        """([B, D, L] embeddings, squeezed biases) of the target ids."""
        return self.item_embeddings(targets).permute(0, 2, 1), self.item_biases(targets).squeeze()


class LSTMNet(_EncoderNet):
    """LSTM over the item embeddings; the hidden state after t items represents the sequence up to t
    (sequence/representations.py:147-262)."""

    def __init__(self, num_items, embedding_dim=32, item_embedding_layer=None, sparse=False):
        super(LSTMNet, self).__init__()
        self._make_item_tables(num_items, embedding_dim, item_embedding_layer, sparse)
        self.lstm = nn.LSTM(batch_first=True, input_size=embedding_dim, hidden_size=embedding_dim)

    def user_representation(self, item_sequences):
        """(states before each item [B, D, L], state after the whole sequence [B, D]) (:195-226)."""
        states, _ = self.lstm(self._shifted_embeddings(item_sequences))
        states = states.permute(0, 2, 1)
        return states[:, :, :-1], states[:, :, -1]

    def forward(self, user_representations, targets):
        """bias[target] + <representation, E[target]> per (sequence, step) (:228-262)."""
        emb, bias = self._target_tables(targets)
        return bias + (user_representations * emb.squeeze()).sum(1).squeeze()


class CNNNet(_EncoderNet):
    """Stacked causal (left-padded) dilated convolutions over time with optional residual connections
    (sequence/representations.py:265-459).  Layers are Conv2d modules named cnn_0, cnn_1, ... as in
    the reference."""

    def __init__(self, num_items, embedding_dim=32, kernel_width=3, dilation=1, num_layers=1,
                 nonlinearity='tanh', residual_connections=True, sparse=False, benchmark=True,
                 item_embedding_layer=None):
        super(CNNNet, self).__init__()
        torch.backends.cudnn.benchmark = benchmark  # MIOpen's find mode on ROCm
        self.kernel_width = _as_tuple(kernel_width, num_layers)
        self.dilation = _as_tuple(dilation, num_layers)
        if nonlinearity not in ('tanh', 'relu'):
            raise ValueError('Nonlinearity must be one of (tanh, relu)')
        self.nonlinearity = torch.tanh if nonlinearity == 'tanh' else F.relu
        self.residual_connections = residual_connections
        self._make_item_tables(num_items, embedding_dim, item_embedding_layer, sparse)
        self.cnn_layers = []
        for k, (width, dil) in enumerate(zip(self.kernel_width, self.dilation)):
            layer = nn.Conv2d(embedding_dim, embedding_dim, (width, 1), dilation=(dil, 1))
            self.add_module('cnn_{}'.format(k), layer)
            self.cnn_layers.append(layer)

    def user_representation(self, item_sequences):
        """(:379-424) every layer sees only the past: the input is padded on the left by the layer's
        receptive field (the first layer one more, which is what shifts the output by a step)."""
        emb = self.item_embeddings(item_sequences).permute(0, 2, 1).unsqueeze(3)  # [B, D, L, 1]
        x = None
        for k, layer in enumerate(self.cnn_layers):
            span = self.kernel_width[k] + (self.kernel_width[k] - 1) * (self.dilation[k] - 1)
            if k == 0:
                x = self.nonlinearity(layer(F.pad(emb, (0, 0, span, 0))))
                if self.residual_connections:
                    x = x + F.pad(emb, (0, 0, 1, 0))
            else:
                below = x
                x = self.nonlinearity(layer(F.pad(x, (0, 0, span - 1, 0))))
                if self.residual_connections:
                    x = x + below
        x = x.squeeze(3)
        return x[:, :, :-1], x[:, :, -1]

    def forward(self, user_representations, targets):
        emb, bias = self._target_tables(targets)
        return bias + (user_representations * emb.squeeze()).sum(1).squeeze()


class MixtureLSTMNet(_EncoderNet):
    """Mixture-of-tastes on top of an LSTM: the hidden state is projected into num_mixtures taste
    vectors and as many attention vectors; an item is scored by the attention-weighted mix of the
    tastes (sequence/representations.py:462-596)."""

    def __init__(self, num_items, embedding_dim=32, num_mixtures=4, item_embedding_layer=None, sparse=False):
        super(MixtureLSTMNet, self).__init__()
        self.num_mixtures = num_mixtures
        self._make_item_tables(num_items, embedding_dim, item_embedding_layer, sparse)
        self.lstm = nn.LSTM(batch_first=True, input_size=embedding_dim, hidden_size=embedding_dim)
        self.projection = nn.Conv1d(embedding_dim, embedding_dim * num_mixtures * 2, kernel_size=1)

    def user_representation(self, item_sequences):
        """([B, 2M, D, L], [B, 2M, D, 1]): tastes in [:, :M], attention vectors in [:, M:] (:520-556)."""
        batch_size, sequence_length = item_sequences.size()
        states, _ = self.lstm(self._shifted_embeddings(item_sequences))
        mixed = self.projection(states.permute(0, 2, 1))
        mixed = mixed.view(batch_size, self.num_mixtures * 2, self.embedding_dim, sequence_length + 1)
        return mixed[:, :, :, :-1], mixed[:, :, :, -1:]

    def forward(self, user_representations, targets):
        """(:558-596) softmax over the mixtures of <attention vector, E[target]>, then the weighted
        taste vector's dot product with E[target], plus the item bias."""
        tastes = user_representations[:, :self.num_mixtures, :, :]
        attention = user_representations[:, self.num_mixtures:, :, :]
        emb = self.item_embeddings(targets).permute(0, 2, 1)
        bias = self.item_biases(targets).squeeze()
        logits = (attention * emb.unsqueeze(1).expand_as(tastes)).sum(2)
        weights = F.softmax(logits, 1).unsqueeze(2).expand_as(tastes)
        blended = (weights * tastes).sum(1)
        return bias + (blended * emb).sum(1).squeeze()
