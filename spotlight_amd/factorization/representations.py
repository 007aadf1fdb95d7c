"""BilinearNet (mirrors spotlight/factorization/representations.py:10-91).

A parameter holder: four embedding tables created in the reference's order (so that
state_dict keys, repr and the torch-generator initialisation match).  Scoring, training
and the optimizer update are done by csrc/slk_bilinear.hip directly on these tensors'
storage; `forward` is provided for API parity and runs the same predict kernel.
"""
import torch
import torch.nn as nn

from spotlight_amd.layers import BloomEmbedding, ScaledEmbedding, ZeroEmbedding


class BilinearNet(nn.Module):

    def __init__(self, num_users, num_items, embedding_dim=32, user_embedding_layer=None,
                 item_embedding_layer=None, sparse=False):
        super(BilinearNet, self).__init__()
        self.embedding_dim = embedding_dim
        if user_embedding_layer is not None:
            self.user_embeddings = user_embedding_layer
        else:
            self.user_embeddings = ScaledEmbedding(num_users, embedding_dim, sparse=sparse)
        if item_embedding_layer is not None:
            self.item_embeddings = item_embedding_layer
        else:
            self.item_embeddings = ScaledEmbedding(num_items, embedding_dim, sparse=sparse)
        self.user_biases = ZeroEmbedding(num_users, 1, sparse=sparse)
        self.item_biases = ZeroEmbedding(num_items, 1, sparse=sparse)

    def tables(self):
        """The four fp32 tables in the C ABI's order (include/spotlight_hip.h: slk_tables); a
        BloomEmbedding layer contributes its compressed table."""
        return [self.user_embeddings.weight, self.item_embeddings.weight,
                self.user_biases.weight, self.item_biases.weight]

    def slk_tables(self):
        """slk_tables over this net's storage (with slk_bloom descriptors for bloom layers)."""
        from spotlight_amd import _native
        w = self.tables()
        bloom = [layer.descriptor() if isinstance(layer, BloomEmbedding) else None
                 for layer in (self.user_embeddings, self.item_embeddings)]
        return _native.make_tables([t.data_ptr() for t in w], w[2].shape[0], w[3].shape[0], w[0].shape[1],
                                   user_bloom=bloom[0], item_bloom=bloom[1])

    def forward(self, user_ids, item_ids):
        """score[k] = <U[user_k], V[item_k]> + bu[user_k] + bi[item_k]  (factorization/representations.py:61-91).
        In eval mode or under torch.no_grad() -- prediction -- this is the fused predict kernel.  In training mode with autograd recording (the model's
        autograd route: an optimizer without a fused update) the four gathers go through the layers' own lookups
        (spotlight_amd/embedding.py: gfx950 gather forward, sorted-ownership scatter backward) and the product / sum through
        torch, exactly the reference's expression, so that loss.backward() reaches the tables."""
        from spotlight_amd.factorization import implicit as host
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            user_embedding = self.user_embeddings(user_ids).squeeze()
            item_embedding = self.item_embeddings(item_ids).squeeze()
            user_bias = self.user_biases(user_ids).squeeze()
            item_bias = self.item_biases(item_ids).squeeze()
            return (user_embedding * item_embedding).sum(1) + user_bias + item_bias
        w = self.tables()
        if w[0].device.type != host._model_device().type:
            raise RuntimeError('BilinearNet.forward runs on the HIP device only (no CPU path)')
        users = user_ids.reshape(-1).to(device=w[0].device, dtype=torch.int64).contiguous()
        items = item_ids.reshape(-1).to(device=w[0].device, dtype=torch.int64).contiguous()
        out = torch.empty(items.numel(), dtype=torch.float32, device=w[0].device)
        eng = host._engine_for(w[0].device)
        tables = self.slk_tables()
        eng.bilinear_predict(tables, users.data_ptr(), users.numel(), items.data_ptr(), items.numel(),
                             out.data_ptr(), host._stream_for(w[0].device))
        return out
