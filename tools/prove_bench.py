#!/usr/bin/env python3
"""Times the whole-proof device path on the BASELINE shapes. usage: prove_bench.py <fib_index> <blowup> <queries> <grinding> [check]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
torch.cuda.init()   # (torch's bundled HIP runtime has to initialise before the library's)
from lambdaworks_cairo_prover_amd import api
idx, b, q, g = (int(x) for x in sys.argv[1:5])
t0 = time.time(); run = api.CairoRun.fibonacci(idx); tr = run.main_trace(); t1 = time.time()
print(f"trace: fib({idx}) steps={run.num_steps} n={run.n_rows} cols={run.n_cols} gen {t1-t0:.2f}s", flush=True)
ctx = api.Context()
opt = api.ProofOptions(b, q, 3, g)
for it in range(2):
    t0 = time.time(); proof = ctx.cairo_prove(tr, run.public_inputs_c, opt); t1 = time.time()
    print(f"prove[{it}]: wall {1e3*(t1-t0):.1f} ms, device rounds {['%.1f' % x for x in ctx.last_round_ms()]} ms, proof {len(proof)} bytes", flush=True)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
for it in range(6):
    t0 = time.time(); p2 = ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt); t1 = time.time()
    if os.environ.get("SP_BENCH_RESIDENT_ONLY"):   # (kernel traces of the resident path alone)
        print(f"warm[{it}]: resident {1e3*(t1-t0):.1f} ms, device rounds {['%.1f' % x for x in ctx.last_round_ms()]} ms", flush=True)
        continue
    t2 = time.time(); p3 = ctx.cairo_prove(tr, run.public_inputs_c, opt); t3 = time.time()
    up_rows = ctx.last_upload_stats()
    t4 = time.time(); p4 = ctx.cairo_prove_run(run, opt); t5 = time.time()
    up_run = ctx.last_upload_stats()
    if it >= 3:
        print(f"warm[{it}]: resident {1e3*(t1-t0):.1f} ms   from host rows {1e3*(t3-t2):.1f} ms   from the run's pinned columns {1e3*(t5-t4):.1f} ms"
              f"   (same bytes: {p2 == p3 == p4 == proof})", flush=True)
        if it == 5:
            print("  upload, host rows:", up_rows, flush=True)
            print("  upload, run columns:", up_run, flush=True)
if len(sys.argv) > 5:
    import oracle_lib as O
    t0 = time.time(); ok = O.cairo_verify(proof, run.public_inputs_c, (b, q, 3, g)); print("oracle verify:", ok, f"{time.time()-t0:.1f}s")
