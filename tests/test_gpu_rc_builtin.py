"""Device proofs of VALID range-check-builtin programs (43 + 18 columns, 50 constraints): same bytes as the CPU oracle, both
verifiers accept, and the composition round takes its 2n-point path (the exact trace check with has_rc_builtin = 1 reports a
clean trace) - plus nearly valid variants that must fall back to the whole domain and still give the oracle's bytes."""
import pytest

from lambdaworks_cairo_prover_amd import api
from test_rc_builtin import run_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,options", [("rc_program", (4, 3, 3, 1)), ("rc_loop_20", (4, 5, 3, 2)), ("output_and_rc", (8, 3, 3, 1)),
                                          ("rc_loop_300", (2, 4, 3, 1)), ("rc_loop_300", (16, 6, 3, 3))])
def test_valid_rc_program_proof_bytes(hip_ctx, oracle, name, options):
    run = run_of(name)
    trace = run.main_trace()
    assert trace.shape[1] == 43
    want = oracle.cairo_prove(trace, run.public_inputs_c, options)
    got = hip_ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*options))
    assert got == want
    assert hip_ctx.last_proof_info()["composition_path"] == 1     # clean trace check -> 2n-point composition
    assert oracle.cairo_verify(got, run.public_inputs_c, options)
    assert api.cairo_verify(got, run.public_inputs_c, api.ProofOptions(*options))


@pytest.mark.parametrize("case", ["rc_limb", "rc_value", "padding_row_limb"])
def test_nearly_valid_rc_traces(hip_ctx, oracle, case):
    run = run_of("rc_loop_20")
    trace = run.main_trace().copy()
    options = (4, 5, 3, 2)
    if case == "rc_limb":
        trace[3, 35, 31] ^= 1            # rc_1 of a used row: the decomposition constraint fails there
    elif case == "rc_value":
        trace[0, 42, 31] ^= 1            # rc_value of row 0
    else:
        trace[trace.shape[0] - 1, 34, 31] ^= 1   # rc_0 of the last (padding) row: the builtin constraint has no exemption
    want = oracle.cairo_prove(trace, run.public_inputs_c, options)
    got = hip_ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*options))
    assert got == want
    assert hip_ctx.last_proof_info()["composition_path"] == 3
    assert not oracle.cairo_verify(got, run.public_inputs_c, options)
    assert not api.cairo_verify(got, run.public_inputs_c, api.ProofOptions(*options))
