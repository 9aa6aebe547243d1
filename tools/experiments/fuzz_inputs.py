#!/usr/bin/env python3
"""Ad-hoc hunt for inputs the reference (restated by the CPU oracle) proves and the device refuses or proves differently:
random programs x random proof options x perturbed public inputs x corrupted cells.  Prints every disagreement.
usage: fuzz_inputs.py [cases=200] [seed0=0]"""
import copy, ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cairo_asm as A
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
from test_rc_builtin import run_of

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = refused = 0
with api.Context(device=0) as ctx:
    for seed in range(seed0, seed0 + cases):
        rng = random.Random(seed)
        pick = rng.random()
        if pick < 0.6:
            words, entry = A.random_program(seed, length=rng.randrange(5, 80))
            run = api.CairoRun.from_program(words, entry_pc=entry)
        elif pick < 0.8:
            run = api.CairoRun.fibonacci(rng.choice([1, 2, 3, 5, 10, 40, 100, 300]))
        else:
            run = run_of(rng.choice(["rc_program", "rc_loop_20", "output_and_rc", "rc_loop_300"]))
        trace = run.main_trace().copy()
        n = trace.shape[0]
        blowup = rng.choice([2, 4, 8, 16, 32, 64, 128] if not os.environ.get("SP_FUZZ_SMALL_BLOWUP") else [2, 4, 8])
        options = (blowup, rng.choice([1, 2, 3, 5, 17, 40]), rng.choice([3, 3, 7, 5]), rng.choice([0, 1, 2, 5, 8]))
        pub = run.public_inputs_c
        what = []
        keep = pub
        if rng.random() < 0.35:                       # perturbed public inputs (a copy of the struct; the pointers stay valid through `run`)
            pub = type(keep)()
            ctypes.memmove(ctypes.byref(pub), ctypes.byref(keep), ctypes.sizeof(keep))
            f = rng.choice(["pc_init", "ap_init", "pc_final", "ap_final", "rc_min", "rc_max", "steps"])
            what.append(f)
            if f == "rc_min": pub.range_check_min = rng.randrange(65536)
            elif f == "rc_max": pub.range_check_max = rng.randrange(65536)
            elif f == "steps": pub.num_steps = rng.randrange(1, n + 1)
            else:
                b = getattr(pub, f)
                b[31] ^= 1 << rng.randrange(8)
        for _ in range(rng.choice([0, 0, 1, 1, 2, 5])):
            r, c = rng.randrange(n), rng.randrange(trace.shape[1])
            byte = rng.choice([31, 30, 20, 8, 1, 0])
            trace[r, c, byte] ^= 1 << rng.randrange(3 if byte == 0 else 8)
            what.append(f"cell({r},{c},{byte})")
        if rng.random() < 0.05:
            trace[:] = 0; what.append("all zero")
        if rng.random() < 0.05:
            trace[:, :, :] = trace[0:1, :, :]; what.append("constant rows")
        try:
            want = oracle.cairo_prove(trace, pub, options)
        except Exception as e:
            refused += 1
            continue                                   # the oracle (the reference's behaviour) refuses: nothing to compare
        try:
            got = ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
        except Exception as e:
            bad += 1
            print(f"seed {seed}: device refuses ({e}) options {options} rows {n} x {trace.shape[1]} {what}")
            continue
        if got != want:
            bad += 1
            print(f"seed {seed}: bytes differ, options {options} rows {n} x {trace.shape[1]} {what}")
print(f"{cases} cases, {refused} refused by the oracle, {bad} disagreements")
