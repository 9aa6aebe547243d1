#!/usr/bin/env python3
"""Times batched forward NTTs on HBM-resident data (beyond the 256 MB infinity cache). usage: ntt_batch_bench.py <log_n> <batch> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lambdaworks_cairo_prover_amd import api
k, batch = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << k
g = torch.Generator(device="cpu"); g.manual_seed(1)
x = torch.randint(0, 2**31 - 1, (batch * n, 8), dtype=torch.int32, generator=g)
x[:, 7] &= 0x03FFFFFF
d = x.cuda()
ctx = api.Context()
ctx.ntt_dev(d.data_ptr(), n, batch); ctx.sync()
ts = []
for _ in range(reps):
    ctx.ntt_dev(d.data_ptr(), n, batch); ctx.sync()
    ts.append(ctx.last_kernel_ms())
ms = min(ts)
print(f"log_n={k} batch={batch}: {ms:.3f} ms per batch, "
      f"{batch * n * k / 2 / ms / 1e6:.2f} G butterflies/s, {batch * n * 64 / ms / 1e6:.1f} GB/s per pass-equivalent of 64 B/elem")
ctx.close()
