"""One-off pin of a full-size proof: the CPU oracle proves fibonacci(<fib_index>) with the given options and prints the
sha256 of the proof bytes (minutes and tens of GB at 2^20 rows). usage: oracle_config3.py <fib_index> <blowup> <queries> <grinding>"""
import sys, time, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from lambdaworks_cairo_prover_amd import api
fib, b, q, g = (int(x) for x in sys.argv[1:5])
run = api.CairoRun.fibonacci(fib)
tr = run.main_trace()
print("trace", tr.shape, flush=True)
t0 = time.time()
proof = O.cairo_prove(tr, run.public_inputs_c, (b, q, 3, g))
print("oracle proof", len(proof), hashlib.sha256(proof).hexdigest(), f"{time.time()-t0:.0f}s", flush=True)
