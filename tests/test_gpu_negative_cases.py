"""tests/integration_tests.rs:206-357 with the DEVICE prover: the proof bytes equal the oracle's (also for the corrupted
traces: the reference proves them, the proof just does not verify) and both verifiers reject."""
import pytest

import negative_cases
from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["slightly_different_program", "range_check_min_plus_one", "range_check_max_minus_one",
                                  "changed_range_check_value", "overflowing_range_check_value", "changed_output",
                                  "different_security_params"])
def test_device_proof_rejected(hip_ctx, oracle, name):
    trace, pub_p, opt_p, pub_v, opt_v, keep = negative_cases.cases()[name]
    proof = hip_ctx.cairo_prove(trace, pub_p, api.ProofOptions(*opt_p))
    assert proof == oracle.cairo_prove(trace, pub_p, opt_p)
    assert not oracle.cairo_verify(proof, pub_v, opt_v)
    assert not api.cairo_verify(proof, pub_v, api.ProofOptions(*opt_v))
