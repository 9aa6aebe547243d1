#!/usr/bin/env python3
"""Host-buffer proof time against the resident one for several gather-thread counts, interleaved in one process (box-to-box
and run-to-run differences are larger than the effect).  usage: host_path_sweep.py [fib=149000] [blowup=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
ctx = api.Context()
for threads in (8, 4, 16, 32, 8, 2, 16):
    ctx.set_option(api.SP_OPT_UPLOAD_THREADS, threads)
    for _ in range(3):
        ctx.cairo_prove(tr, run.public_inputs_c, opt)
    res, host = [], []
    for _ in range(5):
        t0 = time.time(); ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt); res.append(1e3 * (time.time() - t0))
        t0 = time.time(); ctx.cairo_prove(tr, run.public_inputs_c, opt); host.append(1e3 * (time.time() - t0))
    print(f"threads {threads:2d}: resident {min(res):6.1f} ms (median {sorted(res)[2]:6.1f})   host buffer {min(host):6.1f} ms (median {sorted(host)[2]:6.1f})", flush=True)
