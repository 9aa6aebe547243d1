"""Differential test on RANDOM (constraint-violating) traces: the reference proves any trace (the proof just does not
verify), so the device prover must emit the same bytes as the oracle for them too. This exercises every column of the
composition kernel with unstructured values, the general (deg H >= 2n) composition split, and the 61-column / 50-constraint
range-check-builtin layout (reference src/cairo/air.rs:623-629,1141-1160) that no fibonacci trace reaches."""
import random

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
P = api.P


def random_trace(rng, n, cols):
    rows = []
    for _ in range(n):
        row = [rng.randrange(P) for _ in range(cols)]
        for c in (19, 20, 21, 22):           # memory addresses: small integers (sorted as 64-bit keys on the device)
            row[c] = rng.randrange(1, 1 << 20)
        for c in (27, 28, 29):               # instruction offsets: 16-bit
            row[c] = rng.randrange(0, 1 << 16)
        rows.append(row)
    flat = [v for row in rows for v in row]
    return api.felts_to_bytes(flat).reshape(n, cols, 32)


@pytest.mark.parametrize("n,has_rc,options,seed", [(64, False, (4, 3, 3, 1), 1), (128, False, (8, 4, 3, 2), 2), (64, True, (4, 3, 3, 1), 3),
                                                   (256, True, (2, 5, 3, 3), 4), (32, False, (16, 2, 5, 0), 5)])
def test_random_trace_proof_bytes_equal_oracle(hip_ctx, oracle, n, has_rc, options, seed):
    rng = random.Random(seed)
    cols = 43 if has_rc else 34
    trace = random_trace(rng, n, cols)
    pm = [(a, rng.randrange(P)) for a in range(1, 6)]
    segments = [(0, 1000, 1010)] if has_rc else []
    pub, keep = oracle.make_public_inputs(rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100),
                                          rng.randrange(1, 100), 5, 65000, pm, n - 7, segments)
    want = oracle.cairo_prove(trace, pub, options)
    got = hip_ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
    assert len(got) == len(want)
    assert got == want
    assert not oracle.cairo_verify(got, pub, options)  # a random trace does not satisfy the AIR


@pytest.mark.parametrize("n", [2, 4, 8, 16])
@pytest.mark.parametrize("has_rc", [False, True])
def test_tiny_traces(hip_ctx, oracle, n, has_rc):
    """2 .. 16 rows - below anything a Cairo run produces (32 rows) but inside what the ABI admits (a power of two >= 2): transforms of
    length 2, trees of a handful of leaves, FRI with one or two layers, public memory that fills most of the table - the oracle's bytes."""
    cols = 43 if has_rc else 34
    for blowup in (2, 4, 16):
        rng = random.Random(n * 100 + blowup + has_rc)
        trace = random_trace(rng, n, cols)
        pm = [(a, rng.randrange(P)) for a in range(1, min(5, max(1, 4 * n - 3)) + 1)]
        pub, keep = oracle.make_public_inputs(1, 2, 3, 4, 5, 5, 65000, pm, max(1, n - 1), [(0, 1000, 1002)] if has_rc else [])
        options = (blowup, 3, 3, 1)
        assert hip_ctx.cairo_prove(trace, pub, api.ProofOptions(*options)) == oracle.cairo_prove(trace, pub, options), (n, has_rc, blowup)


def _pub_from_run(oracle, run, **override):
    pi = run.public_inputs_c
    kw = dict(pc_init=int.from_bytes(bytes(pi.pc_init), "big"), ap_init=int.from_bytes(bytes(pi.ap_init), "big"),
              fp_init=int.from_bytes(bytes(pi.fp_init), "big"), pc_final=int.from_bytes(bytes(pi.pc_final), "big"),
              ap_final=int.from_bytes(bytes(pi.ap_final), "big"))
    kw.update(override)
    return oracle.make_public_inputs(kw["pc_init"], kw["ap_init"], kw["fp_init"], kw["pc_final"], kw["ap_final"],
                                     pi.range_check_min, pi.range_check_max, run.public_memory(), pi.num_steps)


@pytest.mark.parametrize("case", ["one_cell", "last_row_exempt_cell", "boundary_only", "selector_row"])
def test_nearly_valid_traces_take_the_exact_path(hip_ctx, oracle, case):
    """The composition round decides between its 2n-point path and the whole-domain path with an exact constraint check
    of the trace; traces that are valid except for one detail must still give the oracle's bytes (and an invalid proof)."""
    run = api.CairoRun.fibonacci(40)
    trace = run.main_trace().copy()
    n = trace.shape[0]
    options = (4, 5, 3, 2)
    pub, keep = _pub_from_run(oracle, run)
    if case == "one_cell":
        trace[5, 24, 31] ^= 1                       # dst of step 5: breaks a handful of constraints on one row
    elif case == "last_row_exempt_cell":
        trace[n - 1, 17, 31] ^= 1                   # ap of the last row: only non-exempt constraints can notice
    elif case == "boundary_only":
        pub, keep = _pub_from_run(oracle, run, pc_final=int.from_bytes(bytes(run.public_inputs_c.pc_final), "big") + 1)
    elif case == "selector_row":
        trace[n - 2, 16, 31] ^= 1                   # res on a padding row
    want = oracle.cairo_prove(trace, pub, options)
    got = hip_ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
    assert got == want
    if case in ("one_cell", "boundary_only"):   # (cells of selector-free padding rows are not constrained: those stay valid)
        assert not oracle.cairo_verify(got, pub, options)


def test_valid_trace_still_verifies_after_the_sub_coset_path(hip_ctx, oracle):
    run = api.CairoRun.fibonacci(40)
    options = (4, 5, 3, 2)
    pub, keep = _pub_from_run(oracle, run)
    got = hip_ctx.cairo_prove(run.main_trace(), pub, api.ProofOptions(*options))
    assert got == oracle.cairo_prove(run.main_trace(), pub, options)
    assert oracle.cairo_verify(got, pub, options)


def test_43_column_table_through_every_host_path(hip_ctx):
    """A 2^16-row x 43-column (range-check-builtin layout) random table - 92 MB, above the threshold of the upload pipeline, an
    odd number of columns (the last group of the row-major path is three columns wide, the last column pair of a row shares its
    64-byte line with nothing) - gives the same bytes from the row-major host table, from host columns (ABI encoding, pageable)
    and from device memory.  The small-trace tests above pin those bytes to the oracle; this one pins the big-trace paths to
    each other."""
    import ctypes
    import oracle_lib as oracle
    n, cols = 1 << 16, 43
    rng = np.random.default_rng(43)
    trace = rng.integers(0, 256, size=(n, cols, 32), dtype=np.uint8)
    trace[:, :, 0] &= 0x07                                   # < 2^251 < p
    trace[:, 19:23, :24] = 0                                 # memory addresses: 64-bit
    trace[:, 27:30, :30] = 0                                 # instruction offsets: 16-bit
    prng = random.Random(43)
    pm = [(a, prng.randrange(P)) for a in range(1, 6)]
    pub, keep = oracle.make_public_inputs(3, 5, 5, 7, 9, 5, 65000, pm, n - 7, [(0, 1000, 1010)])
    opt = api.ProofOptions(4, 6, 3, 4)
    rows = hip_ctx.cairo_prove(trace, pub, opt)
    assert hip_ctx.last_upload_stats()["kind"].startswith("row-major")
    assert hip_ctx.last_proof_info()["composition_path"] == 3          # a random trace: deg H >= 2n
    cols_be = np.ascontiguousarray(trace.transpose(1, 0, 2))
    assert hip_ctx.cairo_prove_columns(cols_be, n, cols, pub, opt) == rows
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), trace.nbytes) == 0
    try:
        assert hip.hipMemcpy(dev, trace.ctypes.data, trace.nbytes, 1) == 0
        assert hip_ctx.cairo_prove_dev(dev.value, n, cols, pub, opt) == rows
    finally:
        hip.hipFree(dev)
    assert not api.cairo_verify(rows, pub, opt)
