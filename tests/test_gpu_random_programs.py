"""Random hint-free Cairo programs (tests/cairo_asm.py: random_program) through the device: the main trace built by csrc/trace_kernels.hip
equals the host builder's cell for cell, and the proof - from the run (device-built trace) and from the row-major host table - is the CPU
oracle's, byte for byte."""
import numpy as np
import pytest

import cairo_asm as A
from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(100, 124))
def test_random_program_on_the_device(hip_ctx, oracle, seed):
    words, entry = A.random_program(seed, length=15 + 5 * (seed % 12))
    run = api.CairoRun.from_program(words, entry_pc=entry)
    host = run.main_trace()
    dev = run.main_trace_dev(hip_ctx)
    assert np.array_equal(host, dev), f"device-built trace differs in {int((host != dev).any(axis=2).sum())} cells"
    options = [(4, 3, 3, 1), (2, 5, 3, 2), (8, 4, 3, 1)][seed % 3]
    want = oracle.cairo_prove(host, run.public_inputs_c, options)
    assert hip_ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want
    assert hip_ctx.cairo_prove(host, run.public_inputs_c, api.ProofOptions(*options)) == want
    assert hip_ctx.last_proof_info()["composition_path"] == 1        # a valid trace: the exact check passes, 2n-point composition


@pytest.mark.parametrize("seed", range(200, 232))
def test_random_program_with_one_corrupted_cell(hip_ctx, oracle, seed):
    """One cell of the table changed at random (any column, any row, padding included): the reference still emits a proof for such a
    trace (it does not verify); the device takes whatever composition path its exact trace check leaves and gives the same bytes.
    Every other seed aims at an address column (19 .. 22) or an offset column (27 .. 29) and at a high byte: addresses beyond 2^64
    take the four-limb sort of the auxiliary trace (the reference sorts by the 256-bit value, air.rs:519-523), offsets beyond 2^16
    enter the range-check sort as their low 16 bits (air.rs:689-692)."""
    import random
    rng = random.Random(seed)
    words, entry = A.random_program(seed, length=15 + 5 * (seed % 8))
    run = api.CairoRun.from_program(words, entry_pc=entry)
    trace = run.main_trace().copy()
    r, c = rng.randrange(trace.shape[0]), rng.randrange(trace.shape[1])
    byte = rng.choice([31, 31, 30, 16, 1])
    if seed % 2:
        c, byte = rng.choice([19, 20, 21, 22, 27, 28, 29]), rng.choice([29, 23, 22, 15, 8, 1, 0])
    trace[r, c, byte] ^= 1 << rng.randrange(3 if byte == 0 else 8)          # (byte 0: the value stays below p)
    options = [(4, 3, 3, 1), (2, 5, 3, 2), (8, 4, 3, 1)][seed % 3]
    want = oracle.cairo_prove(trace, run.public_inputs_c, options)
    got = hip_ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*options))
    assert got == want, (r, c, byte)
    cols = np.ascontiguousarray(trace.transpose(1, 0, 2))                   # the same without the presort's early flag (host columns, then the run's round API order)
    assert hip_ctx.cairo_prove_columns(cols, trace.shape[0], trace.shape[1], run.public_inputs_c, api.ProofOptions(*options)) == want
    assert oracle.cairo_verify(got, run.public_inputs_c, options) == api.cairo_verify(got, run.public_inputs_c, api.ProofOptions(*options))


def test_several_huge_addresses_sort_like_the_reference(hip_ctx, oracle):
    """Addresses of every width at once - 2^64, 2^128 + small, 2^192 - 1, values that differ only in their top limb, equal huge values in
    different rows (stability) - in all four address columns: the auxiliary trace's sorted columns and permutation arguments, hence the
    proof, are the oracle's."""
    run = api.CairoRun.fibonacci(40)
    trace = run.main_trace().copy()
    n = trace.shape[0]
    big = [1 << 64, (1 << 128) + 5, (1 << 192) - 1, (1 << 250) + 3, (1 << 250) + 2, (1 << 64) + (1 << 130), 1 << 64, (1 << 250) + 3]
    for k, v in enumerate(big):
        trace[(7 * k + 3) % n, 19 + k % 4] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)
    trace[n - 1, 27] = np.frombuffer(((1 << 16) + 7).to_bytes(32, "big"), dtype=np.uint8)      # an offset beyond 16 bits
    for options in ((4, 3, 3, 1), (2, 4, 3, 2)):
        want = oracle.cairo_prove(trace, run.public_inputs_c, options)
        assert hip_ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*options)) == want
