"""A shape that does not fit the device is refused with SP_E_ALLOC - no abort, no half-set-up prover: the same context proves the next,
smaller shape and gives the oracle's bytes (the reference's `prove` would die in its allocator; include/stark252_hip.h promises an error code)."""
import ctypes

import pytest

from lambdaworks_cairo_prover_amd import _lib, api

pytestmark = pytest.mark.gpu


def test_setup_beyond_device_memory_is_an_error_and_the_context_survives(hip_lib, oracle):
    run = api.CairoRun.fibonacci(100)
    options = (4, 3, 3, 1)
    want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    with api.Context(device=0) as ctx:
        assert ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want
        # 2^25 rows x 52 columns at blowup 16: 0.9 TB of LDE columns alone - more than any single device holds
        opt = api.ProofOptions(16, 80, 3, 20).to_c()
        rc = hip_lib.sp_prove_setup(ctx._h, ctypes.c_uint64(1 << 25), 34, 18, 0, ctypes.byref(opt))
        assert rc == _lib.SP_E_ALLOC, (rc, api.last_error())
        assert "hipMalloc failed" in api.last_error()
        assert ctx.prover_device_bytes() < 1 << 30                      # nothing of the refused shape is left allocated
        assert ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want
        # and through a whole-proof entry point: the error comes back as an exception of the binding, the context still works
        big = api.ProofOptions(128, 3, 3, 1)
        run20 = api.CairoRun.fibonacci(149000)                          # 2^20 rows x blowup 128 = 2^27 points x 52 columns: 223 GB of LDE + the rest
        try:
            proof = ctx.cairo_prove_run(run20, big)
            assert api.cairo_verify(proof, run20.public_inputs_c, big)  # (a device with room for it: then it must be a valid proof)
        except api.SpError as e:
            assert e.code == _lib.SP_E_ALLOC
        assert ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want
