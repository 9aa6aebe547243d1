"""The optional Poseidon Merkle backend (BASELINE.json configs[4]; SP_OPT_MERKLE_BACKEND) on the device: trees and whole proofs with
Poseidon commitments give the bytes of the CPU oracle switched to the same backend, and both verifiers accept them.  The reference
has no such backend (tests/test_poseidon.py says what pins the hash instead)."""
import random

import numpy as np
import pytest

import oracle_lib as O
import poseidon_ref as pr
from lambdaworks_cairo_prover_amd import air, api

pytestmark = pytest.mark.gpu


@pytest.fixture()
def poseidon_ctx(hip_ctx, oracle):
    hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
    oracle.set_merkle_backend(1)
    try:
        yield hip_ctx
    finally:
        oracle.set_merkle_backend(0)
        hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_KECCAK256)


def _rows(rng, n, w, edge=False):
    vals = [[(rng.choice([0, 1, pr.P - 1, 2**251]) if edge and rng.random() < 0.3 else rng.randrange(pr.P)) for _ in range(w)] for _ in range(n)]
    return vals, np.frombuffer(b"".join(int(x).to_bytes(32, "big") for r in vals for x in r), dtype=np.uint8).reshape(n, w, 32)


@pytest.mark.parametrize("n,w", [(1, 1), (1, 4), (2, 1), (2, 2), (8, 3), (64, 1), (64, 18), (256, 34), (512, 43), (4096, 2), (4096, 1)])
def test_device_trees(poseidon_ctx, oracle, n, w):
    rng = random.Random(1000 * n + w)
    vals, rows = _rows(rng, n, w, edge=True)
    root, nodes = poseidon_ctx.merkle_build(rows, want_nodes=True)
    want_root, want_nodes = oracle.merkle_build(rows, want_nodes=True)
    assert root == want_root
    assert np.array_equal(nodes, want_nodes)
    if n <= 64:   # and the plain Python statement of the hash
        leaves = [pr.hash_single(r[0]) if w == 1 else pr.hash_many(r) for r in vals]
        assert int.from_bytes(root, "big") == pr.merkle_root(leaves)


def test_keccak_is_untouched_by_the_switch(hip_ctx, oracle):
    rng = random.Random(5)
    _, rows = _rows(rng, 64, 3)
    before = hip_ctx.merkle_build(rows)
    hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
    try:
        assert hip_ctx.merkle_build(rows) != before
    finally:
        hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_KECCAK256)
    assert hip_ctx.merkle_build(rows) == before == oracle.merkle_build(rows)
    with pytest.raises(Exception):
        hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, 7)


@pytest.mark.parametrize("fib_index,options", [(20, (4, 3, 3, 1)), (100, (8, 5, 3, 2)), (300, (2, 4, 7, 0)), (1000, (4, 6, 3, 3))])
def test_cairo_proof_bytes_equal_oracle(poseidon_ctx, oracle, fib_index, options):
    run = api.CairoRun.fibonacci(fib_index)
    want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    got = poseidon_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    assert got == want
    assert poseidon_ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want      # the column-major host path
    assert oracle.cairo_verify(got, run.public_inputs_c, options)
    assert api.cairo_verify(got, run.public_inputs_c, api.ProofOptions(*options), api.SP_MERKLE_POSEIDON)
    assert not api.cairo_verify(got, run.public_inputs_c, api.ProofOptions(*options))


def test_violating_trace_and_rc_builtin_layout(poseidon_ctx, oracle):
    """The whole-domain composition path and the 43-column layout under Poseidon commitments."""
    from test_gpu_random_traces import random_trace
    rng = random.Random(77)
    for cols, segments in ((34, []), (43, [(0, 1000, 1010)])):
        n = 128
        trace = random_trace(rng, n, cols)
        pm = [(a, rng.randrange(api.P)) for a in range(1, 6)]
        pub, keep = oracle.make_public_inputs(3, 4, 5, 6, 7, 5, 65000, pm, n - 7, segments)
        want = oracle.cairo_prove(trace, pub, (4, 3, 3, 1))
        assert poseidon_ctx.cairo_prove(trace, pub, api.ProofOptions(4, 3, 3, 1)) == want


@pytest.mark.parametrize("kind,length,params,builder", [
    ("simple_fibonacci", 64, (3, 5), lambda n, L: air.simple_fibonacci(3, 5)),      # ONE trace column: still a tree over rows (hash_many)
    ("fibonacci_2_columns", 256, (1, 1), lambda n, L: air.fibonacci_2_columns(1, 1)),
    ("fibonacci_rap", 100, (1, 1), lambda n, L: air.fibonacci_rap(n, L)),
])
def test_example_airs(poseidon_ctx, oracle, kind, length, params, builder):
    options = (4, 3, 3, 1)
    trace = O.example_trace(kind, length, params)
    steps = length if kind == "fibonacci_rap" else 0
    want = O.example_prove(kind, trace, options, params, steps)
    desc, keep = builder(trace.shape[0], length).build()
    got = poseidon_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert got == want
    assert O.example_verify(kind, got, options, params, steps)
    assert api.air_verify(got, desc, api.ProofOptions(*options), api.SP_MERKLE_POSEIDON)
    assert not api.air_verify(got, desc, api.ProofOptions(*options))


def test_mid_size_proof_verifies(poseidon_ctx, oracle):
    """2^16 trace rows, blowup 4 (4.5e6 permutations in the main commitment alone): accepted by both verifiers, a flipped byte is not."""
    run = api.CairoRun.fibonacci(8000)
    opt = api.ProofOptions(4, 20, 3, 8)
    proof = poseidon_ctx.cairo_prove_run(run, opt)
    assert api.cairo_verify(proof, run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON)
    assert oracle.cairo_verify(proof, run.public_inputs_c, (4, 20, 3, 8))
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    assert not api.cairo_verify(bytes(bad), run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON)
