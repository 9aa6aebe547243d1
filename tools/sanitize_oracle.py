#!/usr/bin/env python3
"""Paths of the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize_host.sh builds the instrumented library and preloads the runtimes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib
oracle_lib.LIB_PATH = os.environ.get('SP_SANITIZE_DIR', '/tmp/sp_sanitize') + '/liboracle_stark252.so'
import numpy as np, random
O = oracle_lib
O.load()
import poseidon_ref as pr
rnd = random.Random(1)
# poseidon paths
for n in (1, 2, 3, 18, 34):
    v = [rnd.randrange(pr.P) for _ in range(n)]
    assert O.poseidon(0, v) == pr.hash_many(v)
assert O.poseidon(3, [0, 0, 0]) == pr.hades([0, 0, 0])
# NTT / LDE / merkle both backends
x = np.frombuffer(b''.join(rnd.randrange(pr.P).to_bytes(32, 'big') for _ in range(256)), dtype=np.uint8).reshape(256, 32)
y = O.ntt(x); assert np.array_equal(O.ntt(y, True), x)
O.lde(x, 4, 3)
rows = x.reshape(64, 4, 32)
r0 = O.merkle_build(rows)
O.set_merkle_backend(1); r1 = O.merkle_build(rows, want_nodes=True); O.set_merkle_backend(0)
assert r0 != r1[0]
# whole proofs, both backends, and verification incl. tampered bytes
from lambdaworks_cairo_prover_amd import api
run = api.CairoRun.fibonacci(12)
opts = (4, 3, 3, 1)
for backend in (0, 1):
    O.set_merkle_backend(backend)
    p = O.cairo_prove(run.main_trace(), run.public_inputs_c, opts)
    assert O.cairo_verify(p, run.public_inputs_c, opts)
    for _ in range(30):
        bad = bytearray(p); i = rnd.randrange(len(bad)); bad[i] ^= 1 << rnd.randrange(8)
        O.cairo_verify(bytes(bad), run.public_inputs_c, opts)
    for cut in (0, 5, 8, 100, len(p) - 1):
        O.cairo_verify(p[:cut] or b'\0', run.public_inputs_c, opts)
O.set_merkle_backend(0)
print("oracle under ASan/UBSan: ok")
