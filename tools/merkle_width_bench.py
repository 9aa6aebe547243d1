#!/usr/bin/env python3
"""Times sp_merkle_build_dev for the row widths of the prover (2^22 leaves).  usage: merkle_width_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
n = 1 << 22
for cols in (1, 2, 18, 34, 43, 35):
    data = torch.randint(0, 2**31 - 1, (cols, n, 8), dtype=torch.int32, device="cuda")
    data[..., 7] &= 0x07FFFFFF
    nodes = torch.empty((2 * n - 1, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    perms = n * ((32 * cols + 1 + 135) // 136) + (n - 1)
    print(f"width {cols:2d}: {dt * 1e3:7.3f} ms per tree of 2^22 leaves   {perms / dt / 1e9:6.2f} G Keccak-f/s", flush=True)
    del data, nodes
