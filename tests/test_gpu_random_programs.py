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
