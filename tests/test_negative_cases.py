"""tests/integration_tests.rs:206-357 with the CPU oracle as the prover: every case must be rejected by the oracle's verifier
and by the product's CPU verifier (and the untouched variants must verify - the cases are not vacuous)."""
import pytest

import negative_cases
from lambdaworks_cairo_prover_amd import api

CASES = None


def _cases():
    global CASES
    if CASES is None:
        CASES = negative_cases.cases()
    return CASES


@pytest.mark.parametrize("name", ["slightly_different_program", "range_check_min_plus_one", "range_check_max_minus_one",
                                  "changed_range_check_value", "overflowing_range_check_value", "changed_output",
                                  "different_security_params"])
def test_rejected(oracle, hip_lib, name):
    trace, pub_p, opt_p, pub_v, opt_v, keep = _cases()[name]
    proof = oracle.cairo_prove(trace, pub_p, opt_p)
    assert not oracle.cairo_verify(proof, pub_v, opt_v)
    assert not api.cairo_verify(proof, pub_v, api.ProofOptions(*opt_v))


def test_untouched_variants_verify(oracle, hip_lib):
    for name in ("slightly_different_program", "different_security_params"):
        trace, pub_p, opt_p, pub_v, opt_v, keep = _cases()[name]
        proof = oracle.cairo_prove(trace, pub_p, opt_p)
        assert oracle.cairo_verify(proof, pub_p, opt_p)
        assert api.cairo_verify(proof, pub_p, api.ProofOptions(*opt_p))
