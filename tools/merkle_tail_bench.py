#!/usr/bin/env python3
"""Times sp_merkle_build_dev (one-element leaves, the FRI-layer shape) over tree sizes: the small ones are the latency tail of a proof.
usage: merkle_tail_bench.py   (tools/sweep_merkle_lanes.sh runs it from scratch copies of the tree built with other flags)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
for logn in (6, 8, 10, 12, 13, 14, 15, 16, 18, 20):
    n = 1 << logn
    cols = torch.randint(0, 2**31 - 1, (n * 8,), dtype=torch.int32, device="cuda")
    cols[7::8] &= 0x03ffffff
    nodes = torch.empty((2 * n * 32,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    reps = 200 if logn <= 16 else 20
    for _ in range(5):
        ctx.merkle_build_dev(cols.data_ptr(), n, 1, n, nodes.data_ptr())
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.merkle_build_dev(cols.data_ptr(), n, 1, n, nodes.data_ptr())
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"leaves 2^{logn:<2d}  {dt * 1e6:9.1f} us per tree  ({dt * 1e6 / logn:6.1f} us per level)", flush=True)
