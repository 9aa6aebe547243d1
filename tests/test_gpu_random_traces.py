"""Differential test on RANDOM (constraint-violating) traces: the reference proves any trace (the proof just does not
verify), so the device prover must emit the same bytes as the oracle for them too. This exercises every column of the
composition kernel with unstructured values, the general (deg H >= 2n) composition split, and the 61-column / 50-constraint
range-check-builtin layout (reference src/cairo/air.rs:623-629,1141-1160) that no fibonacci trace reaches."""
import random

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
P = api.P


def random_trace(rng, n, cols):
    rows = []
    for _ in range(n):
        row = [rng.randrange(P) for _ in range(cols)]
        for c in (19, 20, 21, 22):           # memory addresses: small integers (sorted as 64-bit keys on the device)
            row[c] = rng.randrange(1, 1 << 20)
        for c in (27, 28, 29):               # instruction offsets: 16-bit
            row[c] = rng.randrange(0, 1 << 16)
        rows.append(row)
    flat = [v for row in rows for v in row]
    return api.felts_to_bytes(flat).reshape(n, cols, 32)


@pytest.mark.parametrize("n,has_rc,options,seed", [(64, False, (4, 3, 3, 1), 1), (128, False, (8, 4, 3, 2), 2), (64, True, (4, 3, 3, 1), 3),
                                                   (256, True, (2, 5, 3, 3), 4), (32, False, (16, 2, 5, 0), 5)])
def test_random_trace_proof_bytes_equal_oracle(hip_ctx, oracle, n, has_rc, options, seed):
    rng = random.Random(seed)
    cols = 43 if has_rc else 34
    trace = random_trace(rng, n, cols)
    pm = [(a, rng.randrange(P)) for a in range(1, 6)]
    segments = [(0, 1000, 1010)] if has_rc else []
    pub, keep = oracle.make_public_inputs(rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100),
                                          rng.randrange(1, 100), 5, 65000, pm, n - 7, segments)
    want = oracle.cairo_prove(trace, pub, options)
    got = hip_ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
    assert len(got) == len(want)
    assert got == want
    assert not oracle.cairo_verify(got, pub, options)  # a random trace does not satisfy the AIR
