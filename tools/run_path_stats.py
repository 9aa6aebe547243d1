#!/usr/bin/env python3
"""sp_cairo_prove_run beside sp_cairo_prove_dev, interleaved, with the upload statistics of the run path.  usage: run_path_stats.py [fib] [blowup] [iters]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = api.Context()
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
t = {"run": [], "dev": []}
for it in range(iters + 5):
    for path in ("run", "dev"):
        t0 = time.perf_counter()
        if path == "run": ctx.cairo_prove_run(run, opt)
        else: ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
        if it >= 5: t[path].append(1e3 * (time.perf_counter() - t0))
        if path == "run": st = ctx.last_upload_stats()
print(f"fib {fib} blowup {b}: run median {statistics.median(t['run']):.2f} min {min(t['run']):.2f}   dev median {statistics.median(t['dev']):.2f} min {min(t['dev']):.2f}   "
      f"difference of medians {statistics.median(t['run']) - statistics.median(t['dev']):+.2f} ms")
print("  ", st)
