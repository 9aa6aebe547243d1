"""End-to-end parity of the device prover: proof BYTES equal the CPU oracle's and the reference's golden file."""
import hashlib
import os
import struct

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def program_words_from_proof_file(path):
    d = open(path, "rb").read()
    plen = struct.unpack(">Q", d[:8])[0]
    proof, pi = d[8:8 + plen], d[8 + plen:]
    p = 8 + 5 * 32
    for _ in range(2):
        p += 3 if pi[p] == 1 else 1
    nseg = struct.unpack(">Q", pi[p:p + 8])[0]
    p += 8 + 17 * nseg
    npm = struct.unpack(">Q", pi[p:p + 8])[0]
    p += 8
    pm = {}
    for _ in range(npm):
        pm[int.from_bytes(pi[p:p + 32], "big")] = int.from_bytes(pi[p + 32:p + 64], "big")
        p += 64
    return proof, [pm[a] for a in sorted(pm)]


@pytest.mark.parametrize("fib_index,options", [(10, (4, 3, 3, 1)), (100, (4, 3, 3, 1)), (140, (4, 3, 3, 1)), (100, (8, 5, 3, 4)),
                                               (60, (2, 4, 3, 2)), (300, (16, 7, 7, 6))])
def test_device_proof_bytes_equal_oracle(hip_ctx, oracle, fib_index, options):
    run = api.CairoRun.fibonacci(fib_index)
    trace = run.main_trace()
    want = oracle.cairo_prove(trace, run.public_inputs_c, options)
    got = hip_ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*options))
    assert len(got) == len(want)
    assert got == want
    assert oracle.cairo_verify(got, run.public_inputs_c, options)


def test_device_proof_equals_reference_golden_70000(hip_ctx):
    """benches/proofs/fibonacci_70000.proof (n = 2^19, blowup 4 — BASELINE config #4's shape): identical bytes."""
    golden, words = program_words_from_proof_file(os.path.join(GOLDEN, "fibonacci_70000.proof"))
    run = api.CairoRun.from_program(words)
    assert run.n_rows == 1 << 19 and run.num_steps == 490009
    got = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions.default_test_options())
    assert hashlib.sha256(got).hexdigest() == "da962bd4513d991c39a0e0cc11cc76d25b9ec405cebcdaaf1449184d4b54cd6b"
    assert got == golden
