#!/usr/bin/env python3
"""Rank 0's share of a `ranks`-way sharded proof on one GPU over the timing-only transport (sp_comm_init_null): the per-rank compute
time of the projection in bench.py, alone - e.g. under rocprofv3 --kernel-trace for a timeline of where a rank's time goes.
usage: project_one_rank.py <ranks> <fib_index> <blowup> [shard_interp=2] [proofs=8]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ranks, fib, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
proofs = int(sys.argv[5]) if len(sys.argv) > 5 else 8
run = api.CairoRun.fibonacci(fib)
tr = run.main_trace()
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
ctx = api.Context()
ctx.init_null(ranks, 0)
ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, mode)
opt = api.ProofOptions(b, 80, 3, 20)
call = lambda: ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
call()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    call()
ts = []
for _ in range(proofs):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"rank 0 of {ranks}, fib {fib}, blowup {b}, interpolation mode {mode}: median {statistics.median(ts):.2f} ms, min {min(ts):.2f} ms, "
      f"rounds {[round(x, 2) for x in ctx.last_round_ms()]}, info {ctx.last_proof_info()}, stats {ctx.comm_stats()}")
