#!/usr/bin/env python3
"""Soak test of the host-buffer paths: many consecutive proofs from the row-major table (the barrier-free upload job with its worker
threads), from the run's columns and from device memory, every proof compared with the first one; prints the spread of the times.
usage: upload_soak.py [fib=70000] [blowup=4] [iterations=150]"""
import hashlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 150
api.host_bind_to_device(0)
ctx = api.Context()
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
want = hashlib.sha256(ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)).hexdigest()
times = {"rows": [], "run": [], "dev": []}
bad = 0
for it in range(iters):
    for path in ("rows", "run", "dev"):
        t0 = time.perf_counter()
        if path == "rows": p = ctx.cairo_prove(tr, run.public_inputs_c, opt)
        elif path == "run": p = ctx.cairo_prove_run(run, opt)
        else: p = ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
        times[path].append(1e3 * (time.perf_counter() - t0))
        bad += hashlib.sha256(p).hexdigest() != want
for path, v in times.items():
    v2 = sorted(v)
    print(f"{path:4s}: {len(v)} proofs  min {v2[0]:.1f}  median {statistics.median(v2):.1f}  p95 {v2[int(0.95 * len(v2))]:.1f}  max {v2[-1]:.1f} ms")
print("proofs differing from the first:", bad)
sys.exit(1 if bad else 0)
