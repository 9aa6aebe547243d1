#!/usr/bin/env python3
"""Ad-hoc: the reference's example AIRs through sp_air_prove with random lengths, options and corrupted cells against the oracle's
hand-written classes.  usage: fuzz_airs.py [cases=300] [seed0=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from lambdaworks_cairo_prover_amd import air, api
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
BUILD = {"simple_fibonacci": lambda n, L, p: air.simple_fibonacci(*p), "fibonacci_2_columns": lambda n, L, p: air.fibonacci_2_columns(*p),
         "quadratic": lambda n, L, p: air.quadratic(p[0]), "fibonacci_rap": lambda n, L, p: air.fibonacci_rap(n, L), "dummy": lambda n, L, p: air.dummy()}
bad = refused = 0
with api.Context(device=0) as ctx:
    for seed in range(seed0, seed0 + cases):
        rng = random.Random(seed)
        kind = rng.choice(list(BUILD))
        length = rng.choice([4, 8, 16, 20, 32, 64, 100, 128, 256, 512, 1000, 2048])
        params = (rng.randrange(1, 50), rng.randrange(1, 50)) if kind in ("simple_fibonacci", "fibonacci_2_columns") else ((rng.randrange(2, 9), 0) if kind == "quadratic" else (1, 1))
        options = (rng.choice([2, 4, 8, 16, 32, 64, 128]), rng.choice([1, 3, 5, 20]), rng.choice([3, 7]), rng.choice([0, 1, 4]))
        try:
            trace = O.example_trace(kind, length, params).copy()
        except Exception:
            continue
        n = trace.shape[0]
        what = []
        for _ in range(rng.choice([0, 0, 1, 2])):
            r, c, b = rng.randrange(n), rng.randrange(trace.shape[1]), rng.choice([31, 30, 8, 0])
            trace[r, c, b] ^= 1 << rng.randrange(3 if b == 0 else 8)
            what.append((r, c, b))
        steps = length if kind == "fibonacci_rap" else 0
        try:
            want = O.example_prove(kind, trace, options, params, steps)
        except Exception:
            refused += 1
            continue
        desc, keep = BUILD[kind](n, length, params).build()
        try:
            got = ctx.air_prove(desc, trace, api.ProofOptions(*options))
        except Exception as e:
            bad += 1
            print(f"seed {seed}: device refuses ({e}) {kind} length {length} rows {n} options {options} {what}")
            continue
        if got != want:
            bad += 1
            print(f"seed {seed}: bytes differ {kind} length {length} rows {n} options {options} {what}")
print(f"{cases} cases, {refused} refused by the oracle, {bad} disagreements")
