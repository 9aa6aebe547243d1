#!/usr/bin/env python3
"""One batched Merkle build (2^23 leaves x 34 elements, the bench's hash leg) and the FRI-shaped trees (one element per leaf),
timed with HIP events.  Run twice, with and without SP_MK_NO_PAIRS=1, to compare one and two tree levels per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
print("SP_MK_NO_PAIRS =", os.environ.get("SP_MK_NO_PAIRS"))
for logn, cols in ((23, 34), (23, 18), (23, 2), (23, 1), (22, 1), (21, 1), (20, 1), (19, 1)):
    n = 1 << logn
    data = torch.randint(0, 2**31 - 1, (cols, n, 8), dtype=torch.int32, device="cuda")
    data[..., 7] &= 0x07FFFFFF
    nodes = torch.empty((2 * n - 1, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(4):
            ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
        best = min(best, ctx.timer_stop() / 4)
    print(f"2^{logn} leaves x {cols:2d}: {best:8.3f} ms", flush=True)
    del data, nodes
