"""The reference's example AIRs (src/starks/example/*.rs) in the oracle: every integration test of the reference that
proves one of them on Stark252 (tests/integration_tests.rs:36-55, 79-113, 174-205) is replayed - prove, verify, reject a
tampered proof - and the PROGRAM form of each AIR (lambdaworks_cairo_prover_amd/air.py, the form the device library
accepts) must give byte-identical proofs to the hand-written class, which pins the program definitions."""
import pytest

import oracle_lib as O
from lambdaworks_cairo_prover_amd import air

DEFAULT = (4, 3, 3, 1)   # ProofOptions::default_test_options

CASES = [
    ("simple_fibonacci", 8, (1, 1), lambda n: air.simple_fibonacci(1, 1)),
    ("fibonacci_2_columns", 16, (1, 1), lambda n: air.fibonacci_2_columns(1, 1)),
    ("quadratic", 4, (3, 0), lambda n: air.quadratic(3)),
    ("fibonacci_rap", 16, (1, 1), lambda n: air.fibonacci_rap(n, 16)),
    ("dummy", 16, (1, 1), lambda n: air.dummy()),
]


@pytest.mark.parametrize("kind,length,params,builder", CASES)
def test_example_air_proves_and_verifies(oracle, kind, length, params, builder):
    trace = O.example_trace(kind, length, params)
    steps = length if kind == "fibonacci_rap" else 0
    proof = O.example_prove(kind, trace, DEFAULT, params, steps)
    assert O.example_verify(kind, proof, DEFAULT, params, steps)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not O.example_verify(kind, bytes(bad), DEFAULT, params, steps)
    # wrong public input (where the AIR has one)
    if kind in ("simple_fibonacci", "fibonacci_2_columns", "quadratic"):
        assert not O.example_verify(kind, proof, DEFAULT, (params[0] + 1, params[1]), steps)


@pytest.mark.parametrize("kind,length,params,builder", CASES)
@pytest.mark.parametrize("options", [DEFAULT, (8, 5, 3, 2), (2, 4, 7, 0)])
def test_program_form_gives_identical_proofs(oracle, kind, length, params, builder, options):
    trace = O.example_trace(kind, length, params)
    n = trace.shape[0]
    steps = length if kind == "fibonacci_rap" else 0
    want = O.example_prove(kind, trace, options, params, steps)
    desc, keep = builder(n).build()
    got = O.program_air_prove(desc, trace, options)
    assert got == want
    assert O.program_air_verify(desc, got, options)


@pytest.mark.parametrize("kind,length", [("simple_fibonacci", 64), ("fibonacci_2_columns", 128), ("dummy", 256), ("fibonacci_rap", 100)])
def test_larger_traces(oracle, kind, length):
    trace = O.example_trace(kind, length)
    steps = length if kind == "fibonacci_rap" else 0
    proof = O.example_prove(kind, trace, DEFAULT, (1, 1), steps)
    assert O.example_verify(kind, proof, DEFAULT, (1, 1), steps)


@pytest.mark.parametrize("kind,length,params,builder", CASES)
def test_product_verifier_on_program_airs(oracle, hip_lib, kind, length, params, builder):
    """sp_air_verify (the library's CPU verifier, no GPU needed) accepts the oracle's proofs of the example AIRs, agrees
    with the oracle's verifier on tampered proofs, and rejects a proof under another AIR's descriptor."""
    import random
    from lambdaworks_cairo_prover_amd import api
    trace = O.example_trace(kind, length, params)
    n = trace.shape[0]
    steps = length if kind == "fibonacci_rap" else 0
    proof = O.example_prove(kind, trace, DEFAULT, params, steps)
    desc, keep = builder(n).build()
    opts = api.ProofOptions(*DEFAULT)
    assert api.air_verify(proof, desc, opts)
    rng = random.Random(5)
    for _ in range(40):
        bad = bytearray(proof)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        assert api.air_verify(bytes(bad), desc, opts) == O.example_verify(kind, bytes(bad), DEFAULT, params, steps)
    assert not api.air_verify(proof[:-1], desc, opts)
    other, keep2 = (air.quadratic(3) if kind != "quadratic" else air.simple_fibonacci(1, 1)).build()
    assert not api.air_verify(proof, other, opts)
