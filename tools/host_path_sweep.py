#!/usr/bin/env python3
"""Host-buffer proof time against the resident one for several gather-thread counts, interleaved in one process (box-to-box
and run-to-run differences are larger than the effect), with the upload statistics of the library (sp_last_upload_stats).
usage: host_path_sweep.py [fib=149000] [blowup=8] [threads,threads,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [24, 16, 32, 48, 64, 96, 24]
ctx = api.Context()
run = api.CairoRun.fibonacci(fib); tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
print(f"SP_UPLOAD_MAXW={os.environ.get('SP_UPLOAD_MAXW', 'default')} SP_UPLOAD_FREE_RUNNING={os.environ.get('SP_UPLOAD_FREE_RUNNING', '-')}")
for threads in counts:
    ctx.set_option(api.SP_OPT_UPLOAD_THREADS, threads)
    for _ in range(3):
        ctx.cairo_prove(tr, run.public_inputs_c, opt)
    res, host, cols, st = [], [], [], None
    for _ in range(5):
        t0 = time.time(); ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt); res.append(1e3 * (time.time() - t0))
        t0 = time.time(); ctx.cairo_prove(tr, run.public_inputs_c, opt); host.append(1e3 * (time.time() - t0))
        s = ctx.last_upload_stats()
        st = s if st is None or s["exposed_ms"] < st["exposed_ms"] else st
        t0 = time.time(); ctx.cairo_prove_run(run, opt); cols.append(1e3 * (time.time() - t0))
    print(f"threads {threads:3d}: resident {min(res):6.1f} ms (median {sorted(res)[2]:6.1f})   host rows {min(host):6.1f} ms (median {sorted(host)[2]:6.1f})"
          f"   run columns {min(cols):6.1f} (median {sorted(cols)[2]:6.1f})   gather {st['gather_gbs']} GB/s dma {st['dma_gbs']} GB/s exposed {st['exposed_ms']} ms"
          f" max stall {st['max_stall_ms']} host {st['host_ms']} ms", flush=True)
