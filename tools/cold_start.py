#!/usr/bin/env python3
"""First proof of a process: fresh context, first call of a whole-proof entry point (the reference CLI proves once per process,
src/main.rs:85-108).  usage: cold_start.py <rows|run|dev|setup+run> [fib=149000] [blowup=8]   (SP_TIMING=1 for the library's own breakdown)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_imp = time.time()
import torch
torch.cuda.init()   # (torch's bundled HIP runtime has to initialise before the library's)
from lambdaworks_cairo_prover_amd import api, _lib
import ctypes
path = sys.argv[1] if len(sys.argv) > 1 else "rows"
fib = int(sys.argv[2]) if len(sys.argv) > 2 else 149000
b = int(sys.argv[3]) if len(sys.argv) > 3 else 8
opt = api.ProofOptions(b, 80, 3, 20)
t0 = time.time(); ctx = api.Context(); t_ctx = time.time() - t0
t0 = time.time(); run = api.CairoRun.fibonacci(fib); t_run = time.time() - t0
tr = run.main_trace() if path in ("rows", "dev") else None
dev = None
if path == "dev":
    dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
t_setup = 0.0
if path == "setup+run":   # the explicit pre-warm: sp_prove_setup before the trace exists
    t0 = time.time()
    o = opt.to_c()
    _lib.check(ctx._lib.sp_prove_setup(ctx._h, ctypes.c_uint64(run.n_rows), 34, 18, 0, ctypes.byref(o)))
    t_setup = time.time() - t0
times = []
for it in range(3):
    t0 = time.time()
    if path == "rows":
        p = ctx.cairo_prove(tr, run.public_inputs_c, opt)
    elif path == "dev":
        p = ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
    else:
        p = ctx.cairo_prove_run(run, opt)
    times.append(1e3 * (time.time() - t0))
print(f"{path}: n=2^{run.n_rows.bit_length() - 1} blowup {b}: context {1e3 * t_ctx:.1f} ms, run built in {t_run:.2f} s, setup {1e3 * t_setup:.1f} ms, "
      f"first proof {times[0]:.1f} ms, second {times[1]:.1f} ms, third {times[2]:.1f} ms, proof {len(p)} bytes", flush=True)
