#!/usr/bin/env python3
"""Ad-hoc: random world sizes, blowup factors, trace lengths and transport knobs through the sharded prover (the ranks share this box's
GPU and exchange through the host-staged gloo hooks of tests/test_gpu_multirank.py) against the oracle's bytes.
usage: fuzz_multirank.py [cases=30] [seed0=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
import test_gpu_multirank as M

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(seed0, seed0 + cases):
        rng = random.Random(seed)
        world = rng.choice([2, 2, 4, 4, 8])
        kind = rng.choice(["fib", "fib", "prog", "prog_cell", "random", "flip"])
        if kind == "fib": case = M.FIB(rng.choice([5, 10, 40, 100, 300]))
        elif kind == "prog": case = M.PROG(seed, length=rng.randrange(10, 90))
        elif kind == "prog_cell": case = M.PROG(seed, cell=(rng.randrange(64), rng.choice([3, 17, 19, 20, 23, 27, 28, 31]), rng.choice([1 << 70, 5, (1 << 16) + 1, 0])), length=rng.randrange(10, 60))
        elif kind == "random": case = M.RND(rng.choice([64, 128, 256, 512]), seed, rc=rng.random() < 0.3)
        else: case = {"kind": "fib_flip", "fib": rng.choice([10, 40, 100]), "row": rng.randrange(32), "col": rng.randrange(34)}
        options = (rng.choice([2, 4, 8, 16, 32]), rng.choice([1, 3, 6]), 3, rng.choice([0, 1, 2]))
        knobs = {}
        if rng.random() < 0.7: knobs["fri_min_log"] = rng.randrange(3, 9)
        if rng.random() < 0.5: knobs["async"] = True
        if rng.random() < 0.3: knobs["alltoall"] = False
        if rng.random() < 0.3 and knobs.get("async"): knobs["async_a2a"] = False
        if rng.random() < 0.5: knobs["shard_interp"] = rng.choice([0, 1])
        if rng.random() < 0.3: knobs["rows_window"] = True
        if rng.random() < 0.15: knobs["prewarm"] = True
        trace, pub, keep = M._inputs(case)
        want = oracle.cairo_prove(trace, pub, options)
        try:
            results = M._run_world(world, case, options, knobs)
            wrong = [r for r in range(world) if results[r][0] != want]
        except Exception as e:
            wrong = [f"exception {repr(e)[:200]}"]
        if wrong:
            bad += 1
            print(f"seed {seed}: world {world} {case} options {options} knobs {knobs}: ranks {wrong}", flush=True)
    print(f"{cases} cases, {bad} disagreements")
