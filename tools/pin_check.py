import sys, os, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
opt8, opt4 = api.ProofOptions(8, 80, 3, 20), api.ProofOptions(4, 80, 3, 20)
for rep in range(2):
    run = api.CairoRun.fibonacci(149000); print("cfg3 run pinned:", run.columns()[3])
    tr = run.main_trace()
    dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
    ts=[]
    for _ in range(4):
        t0=time.time(); ctx.cairo_prove_run(run, opt8); ts.append(round(1e3*(time.time()-t0),1))
    print('  ms', ts)
    print("  ", ctx.last_upload_stats())
    run2 = api.CairoRun.fibonacci(70000); print("cfg4 run pinned (cfg3 run still alive):", run2.columns()[3])
    ts=[]
    for _ in range(4):
        t0=time.time(); ctx.cairo_prove_run(run2, opt4); ts.append(round(1e3*(time.time()-t0),1))
    print('  ms', ts)
    print("  ", ctx.last_upload_stats())
    del run, run2, tr, dev; gc.collect()
run3 = api.CairoRun.fibonacci(70000); print("cfg4 run pinned (alone):", run3.columns()[3])
ts=[]
for _ in range(4):
    t0=time.time(); ctx.cairo_prove_run(run3, opt4); ts.append(round(1e3*(time.time()-t0),1))
print('  ms', ts)
print("  ", ctx.last_upload_stats())
del run3; gc.collect()
run4 = api.CairoRun.fibonacci(149000); print("cfg3 run pinned (after the frees):", run4.columns()[3])
ts=[]
for _ in range(4):
    t0=time.time(); ctx.cairo_prove_run(run4, opt8); ts.append(round(1e3*(time.time()-t0),1))
print('  ms', ts)
print("  ", ctx.last_upload_stats())
