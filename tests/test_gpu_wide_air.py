"""sp_air_prove beyond the reference's examples: the synthetic AIR of wide_air.py (40 constraints, five frame rows, degrees up
to 3, exemption counts 0, 1, 2, caller-supplied auxiliary trace) gives the oracle's proof bytes on the device - valid and
constraint-violating traces."""
import pytest

import oracle_lib as O
import wide_air
from lambdaworks_cairo_prover_amd import api
from test_wide_air import to_bytes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,options", [(16, (4, 3, 3, 1)), (64, (8, 4, 3, 2)), (32, (2, 5, 3, 0)), (1024, (4, 6, 3, 3))])
def test_device_bytes_equal_oracle(hip_ctx, oracle, n, options):
    desc, keep = wide_air.build(n).build()
    trace = to_bytes(wide_air.main_trace(n))
    want = O.program_air_prove(desc, trace, options)
    got = hip_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert got == want
    assert O.program_air_verify(desc, got, options)
    assert api.air_verify(got, desc, api.ProofOptions(*options))


@pytest.mark.parametrize("cell", [(5, 0), (9, 2), (0, 1)])
def test_violating_traces(hip_ctx, oracle, cell):
    n, options = 64, (4, 3, 3, 1)
    rows = wide_air.main_trace(n)
    rows[cell[0]][cell[1]] ^= 1
    desc, keep = wide_air.build(n, rows).build()
    trace = to_bytes(rows)
    want = O.program_air_prove(desc, trace, options)
    got = hip_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert got == want
    assert not O.program_air_verify(desc, got, options)


def test_dead_value_at_full_slot_occupancy(hip_ctx, oracle):
    """64 values alive at once (the device's slot file is exactly full) and, at that point, one value nobody reads: the host must
    drop it from the device program instead of handing it a slot that holds a live value (the proof would silently stop
    verifying - the oracle and the host verifier evaluate the program without slots)."""
    from lambdaworks_cairo_prover_amd import air
    n, options = 32, (4, 3, 3, 1)
    b = air.AirBuilder(1, [0, 1], 1)
    x, y = b.load(0, 0), b.load(1, 0)
    v = [x + y]
    for _ in range(61):
        v.append(v[-1] + y)                  # v_k = x + k y, all 62 kept alive for the sum below: x, y, v_1 .. v_62 = 64 live values
    _dead = x * y                            # never read
    s = v[0]
    for t in v[1:]:
        s = s + t
    b.constraint(y - x - 1, 1, 1)
    b.constraint(s - 62 * x - 1953 * y, 1, 0)     # identically zero: sum_k (x + k y) = 62 x + 1953 y
    b.boundary(0, 0, 5)
    desc, keep = b.build()
    trace = to_bytes([[5 + i] for i in range(n)])
    want = O.program_air_prove(desc, trace, options)
    assert O.program_air_verify(desc, want, options)
    got = hip_ctx.air_prove(desc, trace, api.ProofOptions(*options))
    assert got == want
    assert api.air_verify(got, desc, api.ProofOptions(*options))
