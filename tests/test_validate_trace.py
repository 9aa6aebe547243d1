"""validate_trace in the sense of reference src/starks/debug.rs:13-104, as used by the reference test
check_simple_cairo_trace_evaluates_to_zero (src/cairo/air.rs:1218-1243): every transition constraint vanishes on every
non-exempt row of main||aux for traces produced by the product front-end, for arbitrary RAP challenges."""
import numpy as np
import pytest

import oracle_lib
from lambdaworks_cairo_prover_amd import api
from test_main_trace_golden import CALL_FUNC, SIMPLE

# air.rs:605-616
EXEMPTIONS = [0] * 16 + [0] + [0, 0, 0] + [1, 1, 1, 1, 0, 0] + [0] * 5 + [0, 0, 0, 1] * 3 + [0, 0, 1] + [0, 0, 0]


def _validate(run, rap):
    main = run.main_trace()
    pub = run.public_inputs_c
    aux = oracle_lib.cairo_aux_trace(main, pub, rap)
    full = np.concatenate([main, aux], axis=1)
    n = full.shape[0]
    assert len(EXEMPTIONS) == 49
    zero = bytes(32)
    for i in range(n):
        frame = np.stack([full[i], full[(i + 1) % n]])
        ev = oracle_lib.cairo_transition(frame, rap)
        for c in range(49):
            if i >= n - EXEMPTIONS[c]:
                continue
            assert ev[c].tobytes() == zero, f"constraint {c} does not vanish at row {i}"


@pytest.mark.parametrize("rap", [(3, 5, 7), (2**250 + 12345, 2**249 + 99, 2**200 + 1)])
def test_simple_and_call_func_traces_validate(oracle, hip_lib, rap):
    _validate(api.CairoRun.from_program(SIMPLE, max_steps=64), rap)
    _validate(api.CairoRun.from_program(CALL_FUNC, max_steps=64, entry_pc=3), rap)


def test_fibonacci_trace_validates(oracle, hip_lib):
    _validate(api.CairoRun.fibonacci(20), (2**250 + 11, 2**248 + 13, 2**247 + 17))


def test_corrupted_trace_is_caught(oracle, hip_lib):
    """Flipping one operand makes some constraint non-zero (negative control for the check above)."""
    run = api.CairoRun.from_program(SIMPLE, max_steps=64)
    main = run.main_trace().copy()
    main[1, 24, 31] ^= 1  # dst value of step 1
    rap = (3, 5, 7)
    aux = oracle_lib.cairo_aux_trace(main, run.public_inputs_c, rap)
    full = np.concatenate([main, aux], axis=1)
    n = full.shape[0]
    bad = 0
    for i in range(n):
        ev = oracle_lib.cairo_transition(np.stack([full[i], full[(i + 1) % n]]), rap)
        bad += sum(1 for c in range(49) if i < n - EXEMPTIONS[c] and ev[c].tobytes() != bytes(32))
    assert bad > 0
