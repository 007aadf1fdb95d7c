"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts
(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads; other widths uncalibrated)."""
import torch
dev = torch.device('cuda', 0)
a = torch.ones(1 << 28, device=dev)                 # 1 GiB fp32
torch.cuda.synchronize()
b = a.clone()                                        # streaming copy: 1 GiB read (16 B/lane), 1 GiB write
c = a.sum()                                          # streaming read 1 GiB
ids64 = torch.randint(0, 10_000_000, (1 << 27,), device=dev, dtype=torch.int64)   # 1 GiB of int64
d = ids64.to(torch.int32)                            # 8 B/lane reads, 4 B/lane writes
table = torch.ones(10_000_000, 64, device=dev)       # 2.56 GB
idx = torch.randint(0, 10_000_000, (1 << 20,), device=dev)
e = table.index_select(0, idx)                       # 1M random 256-B rows = 256 MiB gathered
table.index_add_(0, idx, e)                          # random 256-B row RMW (atomics)
torch.cuda.synchronize()
print('done')
