#!/usr/bin/env python3
"""Proof time right after the context's prover changes shape (config #3 <-> config #4): free + allocate of every device buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
r3, r4 = api.CairoRun.fibonacci(149000), api.CairoRun.fibonacci(70000)
o3, o4 = api.ProofOptions(8, 80, 3, 20), api.ProofOptions(4, 80, 3, 20)
for rep in range(3):
    for name, run, opt in (("cfg3", r3, o3), ("cfg4", r4, o4)):
        ts = []
        for _ in range(3):
            t0 = time.time(); ctx.cairo_prove_run(run, opt); ts.append(round(1e3 * (time.time() - t0), 1))
        print(name, ts, flush=True)
