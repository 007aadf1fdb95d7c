"""PoolNet (mirrors spotlight/sequence/representations.py:26-144).

A parameter holder, like factorization.representations.BilinearNet: two tables created in the
reference's order (item_embeddings: ScaledEmbedding with padding_idx=0, item_biases:
ZeroEmbedding with padding_idx=0), so state_dict keys and the torch-generator initialisation
match.  user_representation / forward are provided for API parity (prediction-time use, no
autograd): training's forward, backward and update are the fused kernels of csrc/slk_seq.hip.
The LSTM / CNN / mixture encoders of the reference are dense MIOpen-style work outside this
package's embedding hot path.
"""
import torch
import torch.nn as nn

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
        """Embedding vectors of `ids` (any shape) -> [..., D]; a BloomEmbedding layer sums its hashed rows
        (layers.py:236-242), computed here with plain torch ops for prediction-time API parity."""
        layer = self.item_embeddings
        if not isinstance(layer, BloomEmbedding):
            return layer.weight[ids]
        import numpy as np
        from sklearn.utils import murmurhash3_32
        flat = ids.reshape(-1).cpu().numpy().astype(np.int32)
        rows = np.stack([murmurhash3_32(flat, seed=seed) % layer.compressed_num_embeddings
                         for seed in layer._masks], axis=1).astype(np.int64)
        rows[flat == layer.padding_idx] = 0
        w = layer.weight
        return w[torch.from_numpy(rows).to(w.device)].sum(1).reshape(tuple(ids.shape) + (w.shape[1],))

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
