#!/usr/bin/env python
"""What stock PyTorch-ROCm ops deliver for the same step on the same GPU ("the
reference on the HIP device via stock PyTorch-ROCm -- a useful second baseline, not the target").

A restatement of the reference's minibatch body (spotlight/factorization/implicit.py:229-243 with
sparse=True embeddings and torch.optim.Adagrad on sparse gradients, spotlight/losses.py:82-90) in plain torch
ops on cuda:0: nn.Embedding(sparse=True) gathers, bpr loss, autograd backward (coalescing sparse gradients),
optimizer step, loss.item().  C2 tables (10M users x 1M items, dim 64); negatives drawn with torch.randint on
the device (the reference draws them with numpy on the host and uploads them: excluded here, in its favour).
usage: python scripts/bench_torch_baseline.py [batch] [steps]"""
import json
import sys
import time

import torch
import torch.nn as nn


class Net(nn.Module):
    def __init__(self, U, I, D):
        super().__init__()
        self.ue, self.ie = nn.Embedding(U, D, sparse=True), nn.Embedding(I, D, sparse=True)
        self.ub, self.ib = nn.Embedding(U, 1, sparse=True), nn.Embedding(I, 1, sparse=True)
        for e in (self.ue, self.ie):
            e.weight.data.normal_(0, 1.0 / D)
        for e in (self.ub, self.ib):
            e.weight.data.zero_()

    def forward(self, u, i):
        return (self.ue(u) * self.ie(i)).sum(1) + self.ub(u).squeeze(1) + self.ib(i).squeeze(1)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    U, I, D = 10_000_000, 1_000_000, 64
    dev = torch.device('cuda', 0)
    net = Net(U, I, D).to(dev)
    opt = torch.optim.Adagrad(net.parameters(), lr=1e-2)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)

    def step():
        u = torch.randint(0, U, (B,), device=dev, generator=gen)
        pos = torch.randint(0, I, (B,), device=dev, generator=gen)
        neg = torch.randint(0, I, (B,), device=dev, generator=gen)
        opt.zero_grad()
        loss = (1.0 - torch.sigmoid(net(u, pos) - net(u, neg))).mean()
        loss.backward()
        opt.step()
        return loss.item()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({'what': 'stock PyTorch-ROCm ops (sparse embeddings + sparse Adagrad), same GPU, C2 tables',
                      'batch': B, 'steps': K, 'ms_per_step': dt * 1e3, 'interactions_per_s': B / dt,
                      'final_loss': last, 'torch': torch.__version__}))


if __name__ == '__main__':
    main()
