#!/usr/bin/env python3
"""Ad-hoc: the fine-grained entry points (sp_ntt, sp_lde, sp_merkle_build, sp_batch_inverse, sp_fe_mul) at random sizes - down to one element -
and with random / extreme values against the oracle.  usage: fuzz_fine.py [cases=300] [seed0=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from lambdaworks_cairo_prover_amd import api
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P = api.P
bad = skipped = 0
def vals(rng, n):
    mode = rng.choice(["rand", "rand", "small", "edge"])
    if mode == "rand": v = [rng.randrange(P) for _ in range(n)]
    elif mode == "small": v = [rng.randrange(4) for _ in range(n)]
    else: v = [rng.choice([0, 1, P - 1, P - 2, 2**251, 2**192, 2**64 - 1]) for _ in range(n)]
    return api.felts_to_bytes(v)
with api.Context(device=0) as ctx:
    for seed in range(seed0, seed0 + cases):
        rng = random.Random(seed)
        op = rng.choice(["ntt", "intt", "coset", "lde", "merkle", "inv", "mul"])
        try:
            if op in ("ntt", "intt", "coset"):
                n = 1 << rng.randrange(0, 14)
                a = vals(rng, n)
                coset = api.felts_to_bytes([rng.choice([3, 7, rng.randrange(1, P)])]) if op == "coset" else None
                inv = op == "intt" or (op == "coset" and rng.random() < 0.5)
                try: want = O.ntt(a, inverse=inv, coset=None if coset is None else int.from_bytes(coset.tobytes(), "big"))
                except Exception: skipped += 1; continue
                got = ctx.ntt(a, inverse=inv, coset=coset)
                ok = np.array_equal(got, want); desc = f"{op} n {n} inverse {inv}"
            elif op == "lde":
                n, b, cols = 1 << rng.randrange(0, 11), rng.choice([1, 2, 4, 8, 16]), rng.randrange(1, 5)
                a = vals(rng, n * cols).reshape(cols, n, 32)
                h = rng.choice([3, 7, rng.randrange(1, P)])
                try: want = np.stack([O.lde(a[c], b, h) for c in range(cols)])
                except Exception: skipped += 1; continue
                got = ctx.lde(a, b, api.felts_to_bytes([h]))
                ok = np.array_equal(got, want); desc = f"lde n {n} blowup {b} cols {cols}"
            elif op == "merkle":
                n, w = 1 << rng.randrange(0, 12), rng.choice([1, 2, 3, 4, 5, 17, 18, 34, 43, 52, 61])
                a = vals(rng, n * w).reshape(n, w, 32)
                try: want = O.merkle_build(a, want_nodes=True)
                except Exception: skipped += 1; continue
                got = ctx.merkle_build(a, want_nodes=True)
                ok = got[0] == want[0] and np.array_equal(got[1], want[1]); desc = f"merkle leaves {n} width {w}"
            elif op == "mul":      # the kernels' own product / square against Python integers
                n = rng.choice([1, 2, 63, 64, 65, 1000, 20000])
                a, sq = vals(rng, n), rng.random() < 0.3
                b = None if sq else vals(rng, n)
                x = api.bytes_to_felts(a); y = x if sq else api.bytes_to_felts(b)
                got = api.bytes_to_felts(ctx.fe_mul(a, b))
                ok = got == [u * v % P for u, v in zip(x, y)]; desc = f"fe_mul n {n} square {sq}"
            else:
                n = rng.choice([1, 2, 3, 5, 64, 255, 256, 257, 1000, 4096, 5000, 70000])
                a = api.felts_to_bytes([rng.randrange(1, P) for _ in range(n)])
                want = O.batch_inverse(a)
                got = ctx.batch_inverse(a)
                ok = np.array_equal(got, want); desc = f"batch inverse n {n}"
        except Exception as e:
            bad += 1; print(f"seed {seed}: {op}: device refuses or fails: {repr(e)[:200]}"); continue
        if not ok:
            bad += 1; print(f"seed {seed}: mismatch: {desc}")
print(f"{cases} cases, {skipped} refused by the oracle, {bad} disagreements")
