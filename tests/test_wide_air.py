"""The generalized program-AIR interface (64 transitions, 8 frame rows, caller-supplied build_auxiliary_trace) on the CPU:
the oracle proves and verifies the synthetic 40-constraint AIR of wide_air.py, the product's CPU verifier agrees, and the
constraints really vanish on its trace (negative control: one flipped cell)."""
import numpy as np
import pytest

import oracle_lib as O
import wide_air
from lambdaworks_cairo_prover_amd import api


def to_bytes(rows):
    n, c = len(rows), len(rows[0])
    return api.felts_to_bytes([v for row in rows for v in row]).reshape(n, c, 32)


@pytest.mark.parametrize("n,options", [(16, (4, 3, 3, 1)), (64, (8, 4, 3, 2)), (32, (2, 5, 3, 0))])
def test_oracle_proves_and_both_verifiers_accept(oracle, hip_lib, n, options):
    b = wide_air.build(n)
    assert len(b.degrees) == 40 and len(b.offsets) == 5 and len(b.ops) > 192
    desc, keep = b.build()
    trace = to_bytes(wide_air.main_trace(n))
    proof = O.program_air_prove(desc, trace, options)
    assert O.program_air_verify(desc, proof, options)
    assert api.air_verify(proof, desc, api.ProofOptions(*options))
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 1
    assert not api.air_verify(bytes(bad), desc, api.ProofOptions(*options))


def test_violating_trace_does_not_verify(oracle, hip_lib):
    n, options = 32, (4, 3, 3, 1)
    rows = wide_air.main_trace(n)
    rows[n // 2][0] ^= 1
    desc, keep = wide_air.build(n, rows).build()
    proof = O.program_air_prove(desc, to_bytes(rows), options)
    assert not O.program_air_verify(desc, proof, options)
    assert not api.air_verify(proof, desc, api.ProofOptions(*options))
