#!/usr/bin/env python3
"""The optional Poseidon Merkle backend on the device: permutation rate of the commitment kernels and whole proofs at the BASELINE
shapes with Poseidon commitments (SP_OPT_MERKLE_BACKEND).  usage: poseidon_bench.py [fib_index blowup queries grinding]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
n = 1 << 20
for cols in (1, 2, 18, 34):
    data = torch.randint(0, 2**31 - 1, (cols, n, 8), dtype=torch.int32, device="cuda")
    data[..., 7] &= 0x07FFFFFF
    nodes = torch.empty((2 * n - 1, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    perms = n * (1 if cols == 1 else (cols + 2) // 2) + (n - 1)
    print(f"width {cols:2d}: {dt * 1e3:8.3f} ms per tree of 2^20 leaves   {perms / dt / 1e9:6.3f} G Hades permutations/s   "
          f"{perms * 214 / dt / 1e9:7.1f} G field products/s", flush=True)
    del data, nodes
args = [int(x) for x in sys.argv[1:]]
for k in range(0, len(args) - 3, 4):
    idx, b, q, g = args[k:k + 4]
    run = api.CairoRun.fibonacci(idx)
    opt = api.ProofOptions(b, q, 3, g)
    for backend, name in ((api.SP_MERKLE_KECCAK256, "keccak256"), (api.SP_MERKLE_POSEIDON, "poseidon")):
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, backend)
        best = None
        for it in range(3):
            t0 = time.perf_counter(); proof = ctx.cairo_prove_run(run, opt); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        t0 = time.perf_counter(); ok = api.cairo_verify(proof, run.public_inputs_c, opt, backend); tv = time.perf_counter() - t0
        print(f"fib({idx}) n={run.n_rows} blowup {b} q {q} g {g}  {name:9s}: {best * 1e3:8.1f} ms  rounds {['%.1f' % x for x in ctx.last_round_ms()]}  "
              f"verified {ok} ({tv * 1e3:.0f} ms on the host)", flush=True)
