#!/usr/bin/env python3
"""sp_cairo_prove_run alone (for a kernel trace).  usage: run_only.py [fib] [blowup] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = api.Context()
run = api.CairoRun.fibonacci(fib)
opt = api.ProofOptions(b, 80, 3, 20)
for it in range(iters):
    t0 = time.perf_counter(); ctx.cairo_prove_run(run, opt)
    print(f"[{it}] run {1e3 * (time.perf_counter() - t0):.1f} ms rounds {['%.1f' % x for x in ctx.last_round_ms()[1:]]} {ctx.last_upload_stats()}", flush=True)
