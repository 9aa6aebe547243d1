"""AIRs other than Cairo on the device (sp_air_prove): the reference's example AIRs (src/starks/example/*.rs) in program
form must give the same proof BYTES as the oracle's hand-written classes - three-row frames, two-row frames, a RAP with
an auxiliary column and two distinct exemption counts, composition degree bounds n and 2n - at the sizes of the
reference's integration tests and larger, for valid and for constraint-violating traces."""
import numpy as np
import pytest

import oracle_lib as O
from lambdaworks_cairo_prover_amd import air, api

pytestmark = pytest.mark.gpu

CASES = [
    ("simple_fibonacci", 8, (1, 1), lambda n, L: air.simple_fibonacci(1, 1)),
    ("simple_fibonacci", 64, (3, 5), lambda n, L: air.simple_fibonacci(3, 5)),
    ("fibonacci_2_columns", 16, (1, 1), lambda n, L: air.fibonacci_2_columns(1, 1)),
    ("fibonacci_2_columns", 256, (1, 1), lambda n, L: air.fibonacci_2_columns(1, 1)),
    ("quadratic", 4, (3, 0), lambda n, L: air.quadratic(3)),
    ("quadratic", 32, (5, 0), lambda n, L: air.quadratic(5)),
    ("fibonacci_rap", 16, (1, 1), lambda n, L: air.fibonacci_rap(n, L)),
    ("fibonacci_rap", 100, (1, 1), lambda n, L: air.fibonacci_rap(n, L)),
    ("dummy", 16, (1, 1), lambda n, L: air.dummy()),
    ("dummy", 128, (1, 1), lambda n, L: air.dummy()),
]


@pytest.mark.parametrize("kind,length,params,builder", CASES)
@pytest.mark.parametrize("options", [(4, 3, 3, 1), (8, 5, 3, 2), (2, 4, 7, 0)])
def test_device_proof_bytes_equal_oracle(hip_ctx, oracle, kind, length, params, builder, options):
    trace = O.example_trace(kind, length, params)
    n = trace.shape[0]
    steps = length if kind == "fibonacci_rap" else 0
    want = O.example_prove(kind, trace, options, params, steps)
    desc, keep = builder(n, length).build()
    got = hip_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert len(got) == len(want)
    assert got == want
    assert O.example_verify(kind, got, options, params, steps)
    assert api.air_verify(got, desc, api.ProofOptions(*options))


@pytest.mark.parametrize("kind,length,builder", [("simple_fibonacci", 32, lambda n, L: air.simple_fibonacci(1, 1)),
                                                 ("fibonacci_2_columns", 32, lambda n, L: air.fibonacci_2_columns(1, 1)),
                                                 ("dummy", 32, lambda n, L: air.dummy()),
                                                 ("fibonacci_rap", 20, lambda n, L: air.fibonacci_rap(n, L))])
def test_violating_traces_give_the_oracle_bytes(hip_ctx, oracle, kind, length, builder):
    """One flipped cell: the reference still emits a (non-verifying) proof; the device must take its whole-domain path
    and produce the same bytes."""
    trace = O.example_trace(kind, length).copy()
    # a cell every transition constraint reads (column 0 of the dummy AIR only has to be a bit, and the permuted column of
    # fibonacci_rap is not tied to anything by that example, so those two would stay valid)
    trace[trace.shape[0] // 2, 1 if kind == "dummy" else 0, 31] ^= 1
    n = trace.shape[0]
    steps = length if kind == "fibonacci_rap" else 0
    options = (4, 3, 3, 1)
    want = O.example_prove(kind, trace, options, (1, 1), steps)
    desc, keep = builder(n, length).build()
    got = hip_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert got == want
    assert not O.example_verify(kind, got, options, (1, 1), steps)


def test_malformed_programs_are_rejected(hip_ctx):
    b = air.simple_fibonacci(1, 1)
    b.ops[0] = (air.OP_ADD, 5, 7)   # refers to later values
    desc, keep = b.build()
    trace = O.example_trace("simple_fibonacci", 8)
    with pytest.raises(api.SpError):
        hip_ctx.air_prove(desc, trace, api.ProofOptions(4, 3, 3, 1))
    b2 = air.simple_fibonacci(1, 1)
    b2.ops[0] = (air.OP_LOAD, 0, 9)  # column out of range
    desc2, keep2 = b2.build()
    with pytest.raises(api.SpError):
        hip_ctx.air_prove(desc2, trace, api.ProofOptions(4, 3, 3, 1))
    # after the failures the context still proves
    desc3, keep3 = air.simple_fibonacci(1, 1).build()
    assert hip_ctx.air_prove(desc3, trace, api.ProofOptions(4, 3, 3, 1)) == O.example_prove("simple_fibonacci", trace, (4, 3, 3, 1))
