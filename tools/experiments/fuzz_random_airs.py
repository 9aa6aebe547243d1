#!/usr/bin/env python3
"""Ad-hoc: RANDOM AIRs in program form (random column counts, frame heights, constraint expressions, degrees, exemptions, boundary
constraints) on random traces through sp_air_prove against the oracle's interpreter (oracle_program_air_prove), and the two
verifiers' verdicts on the result.  usage: fuzz_random_airs.py [cases=300] [seed0=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from lambdaworks_cairo_prover_amd import air, api
P = air.P
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = refused = 0
with api.Context(device=0) as ctx:
    for seed in range(seed0, seed0 + cases):
        rng = random.Random(seed)
        big = os.environ.get('SP_FUZZ_BIG_AIRS') is not None
        cols, R, f = rng.randrange(1, 30 if big else 7), rng.randrange(1, 9 if big else 5), rng.choice([1, 2, 2, 3])
        n = 1 << rng.randrange(max(2, R.bit_length() + 1), 9)
        nex = rng.randrange(0, 3)
        b = air.AirBuilder(cols, list(range(R)), f, num_transition_exemptions=max(1, nex))
        def expr(depth):
            if depth == 0 or rng.random() < 0.3:
                return (b.load(rng.randrange(R), rng.randrange(cols)), 1) if rng.random() < 0.75 else (b.const(rng.choice([0, 1, 2, rng.randrange(P)])), 0)
            (x, dx), (y, dy) = expr(depth - 1), expr(depth - 1)
            op = rng.choice("+-*")
            if op == "*" and dx + dy > f + 1: op = "+"
            return (x + y, max(dx, dy)) if op == "+" else (x - y, max(dx, dy)) if op == "-" else (x * y, dx + dy)
        for _ in range(rng.randrange(1, 65 if big else 9)):
            v, d = expr(rng.randrange(1, 5 if big else 4))
            b.constraint(v, max(1, min(d, f + 1)) if rng.random() < 0.8 else rng.randrange(1, f + 2), rng.randrange(0, max(1, nex) + 1))
        for _ in range(rng.randrange(0, 4)):
            b.boundary(rng.randrange(cols), rng.randrange(n), rng.randrange(P))
        trace = api.felts_to_bytes([rng.choice([0, 1, rng.randrange(P)]) for _ in range(n * cols)]).reshape(n, cols, 32)
        options = (rng.choice([2, 4, 8, 16]), rng.choice([1, 3, 7]), rng.choice([3, 7]), rng.choice([0, 1, 3]))
        desc, keep = b.build()
        try:
            want = O.program_air_prove(desc, trace, options)
        except Exception:
            refused += 1
            continue
        try:
            got = ctx.air_prove(desc, trace, api.ProofOptions(*options))
        except Exception as e:
            bad += 1
            print(f"seed {seed}: device refuses ({str(e)[:120]}) cols {cols} rows {n} frame {R} factor {f} options {options}")
            continue
        if got != want:
            bad += 1
            print(f"seed {seed}: bytes differ cols {cols} rows {n} frame {R} factor {f} options {options}")
        elif O.program_air_verify(desc, got, options) != api.air_verify(got, desc, api.ProofOptions(*options)):
            bad += 1
            print(f"seed {seed}: verdicts differ")
print(f"{cases} cases, {refused} refused by the oracle, {bad} disagreements")
