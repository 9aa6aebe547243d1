"""Traces of 2 .. 16 rows (below anything a Cairo run produces: 32 rows) through sp_cairo_prove against the CPU oracle: what the smallest
shapes the ABI admits do.  usage: tiny_traces.py"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
from test_gpu_random_traces import random_trace
bad = 0
with api.Context(device=0) as ctx:
    for n in (2, 4, 8, 16):
        for has_rc in (False, True):
            for blowup in (2, 4, 16):
                rng = random.Random(n * 100 + blowup + has_rc)
                cols = 43 if has_rc else 34
                trace = random_trace(rng, n, cols)
                pm_n = min(5, max(1, 4 * n - 3))
                pm = [(a, rng.randrange(api.P)) for a in range(1, pm_n + 1)]
                segs = [(0, 1000, 1002)] if has_rc else []
                pub, keep = oracle.make_public_inputs(1, 2, 3, 4, 5, 5, 65000, pm, max(1, n - 1), segs)
                options = (blowup, 3, 3, 1)
                try:
                    want = oracle.cairo_prove(trace, pub, options)
                except Exception as e:
                    print("oracle refuses", n, has_rc, blowup, repr(e)[:100]); continue
                try:
                    got = ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
                    ok = got == want
                except api.SpError as e:
                    ok = False; print("device refuses", n, has_rc, blowup, str(e)[:200])
                bad += not ok
                print(n, has_rc, blowup, "ok" if ok else "DIFFERENT")
print("bad", bad)
