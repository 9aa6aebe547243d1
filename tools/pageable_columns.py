#!/usr/bin/env python3
"""sp_cairo_prove_columns from column-major tables in PAGEABLE memory (a foreign column-major TraceTable, e.g. `trace.cols()` of
the reference collected into one Vec), against the page-locked run and the resident call.  usage: pageable_columns.py [fib] [blowup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = api.Context()
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
cols = np.ascontiguousarray(tr.transpose(1, 0, 2))          # (cols, n, 32) canonical big-endian, pageable
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
ref = ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
for it in range(5):
    t0 = time.perf_counter(); ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt); td = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); p = ctx.cairo_prove_columns(cols, tr.shape[0], tr.shape[1], run.public_inputs_c, opt); tc = 1e3 * (time.perf_counter() - t0)
    s = ctx.last_upload_stats()
    t0 = time.perf_counter(); ctx.cairo_prove_run(run, opt); trn = 1e3 * (time.perf_counter() - t0)
    print(f"[{it}] resident {td:6.1f}   pageable columns {tc:6.1f} (same bytes {p == ref})   run {trn:6.1f}   {s}", flush=True)
