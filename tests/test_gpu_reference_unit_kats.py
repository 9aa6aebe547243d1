"""The reference unit-test vectors of tests/test_reference_unit_kats.py through the DEVICE: the main trace built by
csrc/trace_kernels.hip (sp_cairo_run_main_trace_dev: step_rows_kernel, rc_builtin_kernel, tail_rows_kernel) on the crafted runs -
checked against the reference's expected values AND cell for cell against the host builder - and the auxiliary trace of
csrc/aux_kernels.hip (public-memory substitution, sort by address) through sp_cairo_commit_aux's root against the root of the
oracle's auxiliary trace on the inputs of air.rs:1246-1409 (the oracle itself is pinned to those vectors on the CPU)."""
import ctypes

import numpy as np
import pytest

import test_reference_unit_kats as K
from lambdaworks_cairo_prover_amd import _lib, api

pytestmark = pytest.mark.gpu


def both(run, hip_ctx):
    dev = run.main_trace_dev(hip_ctx)
    assert np.array_equal(dev, run.main_trace()), "device-built main trace differs from the host builder's"
    return dev


def test_rc_decompose_device_builder(hip_ctx):
    run, v, values = K.rc_decompose_run()
    K.check_rc_decompose(both(run, hip_ctx), v, values)


def test_fill_range_check_values_device_builder(hip_ctx):
    run, v = K.rc_holes_run()
    K.check_rc_holes(run, both(run, hip_ctx), v)


def test_add_missing_values_to_offsets_column_device_builder(hip_ctx):
    run, v = K.missing_offsets_run()
    K.check_missing_offsets(both(run, hip_ctx), v)


@pytest.mark.parametrize("name", ["no_codelen", "inside_program_section", "outside_program_section"])
def test_get_memory_holes_device_builder(hip_ctx, name):
    run, v = K.memory_holes_run(name)
    K.check_memory_holes(run, both(run, hip_ctx), v)


def test_fill_memory_holes_device_builder(hip_ctx):
    run, v = K.fill_memory_holes_run()
    K.check_fill_memory_holes(run, both(run, hip_ctx), v)


@pytest.mark.parametrize("key", ["add_program", "add_program_with_output", "sort_columns_by_memory_address"])
@pytest.mark.parametrize("n", [32, 1024])
def test_aux_trace_root_on_the_reference_unit_inputs(hip_lib, oracle, key, n):
    """sp_cairo_commit_aux on a main trace whose LAST rows hold the reference test's (a, v) and whose other rows are (1, 1) accesses
    (the public-memory section is the tail of the flattened columns, air.rs:475-494): the commitment of the oracle's auxiliary trace."""
    v = K.vectors()[key]
    a, val = list(v["a"]), list(v["v"])
    pad = 4 * n - len(a)
    a, val = [1] * pad + a, [1] * pad + val
    trace, pub_args = K.aux_inputs(a, val, v.get("public_memory", []), v.get("output_range"))
    pub, keep = oracle.make_public_inputs(*pub_args)
    blowup = 4
    aux = oracle.cairo_aux_trace(trace, pub, K.RAP)                       # (n, 18, 32)
    lde_cols = [oracle.lde(oracle.ntt(np.ascontiguousarray(aux[:, j]), inverse=True), blowup, 3) for j in range(18)]
    rows = np.ascontiguousarray(np.stack(lde_cols, axis=1))               # (N, 18, 32), natural order (prover.rs:144-148)
    want = oracle.merkle_build(rows)
    with api.Context(device=0) as ctx:
        opt = api.ProofOptions(blowup, 3, 3, 1).to_c()
        root = (ctypes.c_uint8 * 32)()
        u8p = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        _lib.check(hip_lib.sp_prove_setup(ctx._h, ctypes.c_uint64(n), 34, 18, 0, ctypes.byref(opt)))
        _lib.check(hip_lib.sp_commit_trace(ctx._h, 0, u8p(trace), ctypes.c_uint64(n), 34, root))
        rap_b = b"".join(int(x).to_bytes(32, "big") for x in K.RAP)
        _lib.check(hip_lib.sp_cairo_commit_aux(ctx._h, rap_b, ctypes.byref(pub), root))
    assert bytes(root) == want


def test_domain_constructor_and_lde_edge_case_on_the_device(hip_ctx, oracle):
    """The same two reference tests (prover.rs:787-835, 865-882) through sp_lde: the device's domain tables (gen_roots_kernel, a18) are
    the reference's `lde_roots_of_unity_coset`."""
    P = api.P
    n, blowup, offset = 8, 2, 3
    w = oracle.primitive_root(4)
    x = api.felts_to_bytes([0, 1] + [0] * (n - 2)).reshape(1, n, 32)
    assert api.bytes_to_felts(hip_ctx.lde(x, blowup, api.felts_to_bytes([offset]))[0]) == [offset * pow(w, i, P) % P for i in range(16)]
    w32 = oracle.primitive_root(5)
    x8 = api.felts_to_bytes([0] * 8 + [1] + [0] * 7).reshape(1, 16, 32)
    assert api.bytes_to_felts(hip_ctx.lde(x8, 2, api.felts_to_bytes([3]))[0]) == [pow(3 * pow(w32, i, P) % P, 8, P) for i in range(32)]
    for k in (10, 16):                                           # and at sizes that go through the LDS-tiled passes: p(x) = x again
        n = 1 << k
        w = oracle.primitive_root(k + 2)
        x = api.felts_to_bytes([0, 1] + [0] * (n - 2)).reshape(1, n, 32)
        got = api.bytes_to_felts(hip_ctx.lde(x, 4, api.felts_to_bytes([7]))[0])
        acc, want = 7, []
        for _ in range(4 * n):
            want.append(acc)
            acc = acc * w % P
        assert got == want
