#!/usr/bin/env python3
"""The row-major host path alone (for a kernel trace of it).  usage: rows_only.py [fib=149000] [blowup=8] [iterations=5] [path=rows|dev|run]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
path = sys.argv[4] if len(sys.argv) > 4 else "rows"
ctx = api.Context()
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
print("load average", os.getloadavg())
for it in range(iters):
    t0 = time.perf_counter()
    if path == "rows":
        ctx.cairo_prove(tr, run.public_inputs_c, opt)
    elif path == "run":
        ctx.cairo_prove_run(run, opt)
    else:
        ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
    print(f"[{it}] {path} {1e3 * (time.perf_counter() - t0):.1f} ms rounds {['%.1f' % x for x in ctx.last_round_ms()[1:]]} {ctx.last_upload_stats() if path != 'dev' else ''}", flush=True)
