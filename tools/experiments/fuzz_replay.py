#!/usr/bin/env python3
"""Random splits of the sharded prover through the replay harness (tools/replay_ranks.py): world sizes 2 .. 32 (beyond the 8 ranks that can
share a GPU concurrently), any blowup factor, blocking or stream-ordered hooks, the three interpolation modes, both Merkle backends, the run or
the row-major host table as input - every rank's bytes against the CPU oracle's.  usage: fuzz_replay.py [cases=40] [seed0=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
torch.cuda.init()
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
from replay_ranks import sharded_proof_by_replay

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(seed0, seed0 + cases):
    rng = random.Random(seed)
    fib = rng.choice([30, 100, 300, 1000, 2000])
    blowup = rng.choice([2, 4, 8, 16, 32])
    world = rng.choice([2, 4, 8, 16, 32])
    run = api.CairoRun.fibonacci(fib)
    n = run.n_rows
    if n * blowup < 2 * min(world, blowup) ** 2 or n * blowup > 1 << 20:
        continue
    options = (blowup, rng.choice([1, 3, 9]), rng.choice([3, 7]), rng.choice([0, 1, 5]))
    opt = api.ProofOptions(*options)
    poseidon, so, mode, rows = rng.random() < 0.25, rng.random() < 0.5, rng.choice([0, 1, 2]), rng.random() < 0.4
    oracle.set_merkle_backend(1 if poseidon else 0)
    try:
        want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    finally:
        oracle.set_merkle_backend(0)
    trace = run.main_trace() if rows else None
    prove = (lambda c: c.cairo_prove(trace, run.public_inputs_c, opt)) if rows else (lambda c: c.cairo_prove_run(run, opt))
    try:
        with api.Context(device=0) as ctx:
            if poseidon:
                ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
            ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, mode)
            ctx.set_option(api.SP_OPT_FRI_SHARD_MIN_LOG, rng.choice([4, 8, 16]))
            proofs, stats = sharded_proof_by_replay(api, ctx, prove, world, log=lambda *_: None, stream_ordered=so)
        ok = sorted(proofs) == list(range(world)) and all(p == want for p in proofs.values())
    except Exception as e:
        ok = False
        print("  exception:", repr(e)[:300])
    if not ok:
        bad += 1
        print(f"DISAGREEMENT seed {seed}: fib {fib} rows {n} options {options} world {world} poseidon {poseidon} stream-ordered {so} mode {mode} rows-entry {rows}")
print(f"{cases} cases, {bad} disagreements")
sys.exit(1 if bad else 0)
