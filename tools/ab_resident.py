#!/usr/bin/env python3
"""Interleaved same-box A/B of the resident proof time under an environment switch the library reads once per process.
usage: ab_resident.py <ENV_VAR> <fib_index> <blowup> [rounds=4] [proofs=12]
Each round runs a fresh child WITHOUT the variable and one WITH it (=1); a child warms for 300 ms and prints the median and the
minimum of `proofs` sp_cairo_prove_dev calls (80 queries, grinding 20) plus the device time of round 4."""
import os, subprocess, sys, statistics, json
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) >= 2 and sys.argv[1] == "--child":
    import time
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    torch.cuda.init()
    from lambdaworks_cairo_prover_amd import api
    fib, b, proofs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    run = api.CairoRun.fibonacci(fib)
    tr = run.main_trace()
    dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
    ctx = api.Context()
    opt = api.ProofOptions(b, 80, 3, 20)
    call = lambda: ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
    call()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        call()
    ts, r4 = [], []
    for _ in range(proofs):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3); r4.append(ctx.last_round_ms()[4])
    print(json.dumps({"median": statistics.median(ts), "min": min(ts), "round4_median": statistics.median(r4)}))
    sys.exit(0)
var, fib, b = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
proofs = sys.argv[5] if len(sys.argv) > 5 else "12"
res = {"without": [], "with": []}
for r in range(rounds):
    for name in ("without", "with"):
        env = dict(os.environ)
        env.pop(var, None)
        if name == "with":
            env[var] = "1"
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", fib, b, proofs], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        res[name].append(json.loads(line))
        print(f"round {r} {name:8s} {var}: {line}", flush=True)
for name in ("without", "with"):
    print(f"{name:8s} {var}: median of medians {statistics.median(x['median'] for x in res[name]):.3f} ms, best min {min(x['min'] for x in res[name]):.3f} ms, "
          f"round 4 {statistics.median(x['round4_median'] for x in res[name]):.3f} ms")
