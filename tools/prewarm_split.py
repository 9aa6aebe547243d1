#!/usr/bin/env python3
"""What sp_prewarm spends its time on: fresh processes, one flag set each (no VM beside it).
usage: prewarm_split.py [log2 rows=19] [blowup=4]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    flags, logn, blowup = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    torch.cuda.init()
    from lambdaworks_cairo_prover_amd import api
    t0 = time.perf_counter(); ctx = api.Context(); t_ctx = 1e3 * (time.perf_counter() - t0)
    opt = api.ProofOptions(blowup, 80, 3, 20)
    out = {"flags": flags, "ctx_ms": round(t_ctx, 1)}
    t0 = time.perf_counter()
    if flags < 0:
        import ctypes
        from lambdaworks_cairo_prover_amd import _lib
        o = opt.to_c()
        _lib.check(_lib.load().sp_prove_setup(ctx._h, ctypes.c_uint64(1 << logn), 34, 18, 0, ctypes.byref(o)))
    else: ctx.prewarm(1 << logn, 34, 18, False, opt, flags)
    out["call_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    t0 = time.perf_counter(); ctx.prewarm(1 << logn, 34, 18, False, opt, 7); out["then_all_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    print(json.dumps(out))
    sys.exit(0)
logn = sys.argv[1] if len(sys.argv) > 1 else "19"
blowup = sys.argv[2] if len(sys.argv) > 2 else "4"
for flags in (-1, 1, 2, 4, 7, 7):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(flags), logn, blowup], capture_output=True, text=True)
    print(f"2^{logn} x blowup {blowup}:", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
