"""The Cairo main trace built ON THE DEVICE (csrc/trace_kernels.hip; sp_cairo_prove_run's default input path) equals the table the
host builder produces - itself pinned by the expected tables of the reference's own unit tests (tests/test_main_trace_golden.py,
src/cairo/execution_trace.rs:660-1161) - cell for cell: on those unit-test programs, on the reference's binary dumps
(tests/golden/program.{trace,memory}, mul_{trace,mem}.out), on fibonacci runs with and without memory holes / range-check holes,
and on the range-check-builtin programs (43 columns).  And every proof-byte pin once through this path."""
import hashlib
import os

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(run, ctx):
    host = run.main_trace()
    dev = run.main_trace_dev(ctx)
    if not np.array_equal(host, dev):
        bad = np.argwhere((host != dev).any(axis=2))
        r, c = bad[0]
        raise AssertionError(f"{len(bad)} cells differ; first at row {r}, column {c}: host {host[r, c].tobytes().hex()} device {dev[r, c].tobytes().hex()} "
                             f"(steps {run.num_steps}, rows {run.n_rows})")
    lw = run.main_trace_dev(ctx, api.SP_FE_MONT_LIMBS)
    assert np.array_equal(lw, run.main_trace(api.SP_FE_MONT_LIMBS))


def test_unit_test_programs_of_the_reference(hip_ctx):
    from test_main_trace_golden import CALL_FUNC, SIMPLE
    _same(api.CairoRun.from_program(SIMPLE, max_steps=64, entry_pc=1), hip_ctx)
    _same(api.CairoRun.from_program(CALL_FUNC, max_steps=64, entry_pc=3), hip_ctx)


def test_binary_dumps_of_the_reference(hip_ctx):
    t = open(os.path.join(GOLDEN, "mul_trace.out"), "rb").read()
    m = open(os.path.join(GOLDEN, "mul_mem.out"), "rb").read()
    _same(api.CairoRun.from_dumps(t, m, program_size=5), hip_ctx)
    t = open(os.path.join(GOLDEN, "program.trace"), "rb").read()
    m = open(os.path.join(GOLDEN, "program.memory"), "rb").read()
    for size in (1, 5, len(m) // 40):           # the size of the public memory moves the first memory-hole address and the padding
        try:
            run = api.CairoRun.from_dumps(t, m, program_size=size)
        except api.SpError:
            continue
        _same(run, hip_ctx)


@pytest.mark.parametrize("fib_index", [1, 2, 10, 35, 100, 1000, 9000])
def test_fibonacci_runs(hip_ctx, fib_index):
    _same(api.CairoRun.fibonacci(fib_index), hip_ctx)


def test_programs_with_jumps_holes_and_builtins(hip_ctx):
    import cairo_asm as A
    from test_rc_builtin import PROGRAMS, run_of
    for name in PROGRAMS:
        run = run_of(name)
        assert run.n_cols == 43
        _same(run, hip_ctx)
    # a taken and a not-taken jnz, memory the run never touches (15 rows of memory holes) and a wide offset range (20 rows of
    # range-check holes) in the plain 34-column layout
    words, entry = A.holes_program()
    run = api.CairoRun.from_program(words, entry_pc=entry)
    assert run.num_steps == 7 and run.n_rows == 64
    _same(run, hip_ctx)


def test_proof_bytes_through_the_device_built_trace(hip_ctx, oracle):
    """sp_cairo_prove_run with the device-built trace (the default), with the option off, and the row-major call: the oracle's bytes."""
    for fib_index, options in [(10, (4, 3, 3, 1)), (140, (4, 3, 3, 1)), (300, (8, 5, 3, 4)), (60, (2, 4, 3, 2))]:
        run = api.CairoRun.fibonacci(fib_index)
        want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
        opt = api.ProofOptions(*options)
        assert hip_ctx.cairo_prove_run(run, opt) == want
        assert hip_ctx.last_upload_stats()["kind"].startswith("run image")
        hip_ctx.set_option(api.SP_OPT_DEVICE_TRACE, 0)
        try:
            assert hip_ctx.cairo_prove_run(run, opt) == want
            assert hip_ctx.last_upload_stats()["kind"].startswith("host columns")
        finally:
            hip_ctx.set_option(api.SP_OPT_DEVICE_TRACE, 1)


def test_golden_70000_and_config3_through_the_device_built_trace(hip_ctx):
    from test_gpu_prover import program_words_from_proof_file
    golden, words = program_words_from_proof_file(os.path.join(GOLDEN, "fibonacci_70000.proof"))
    run = api.CairoRun.from_program(words)
    got = hip_ctx.cairo_prove_run(run, api.ProofOptions.default_test_options())
    assert got == golden
    st = hip_ctx.last_upload_stats()
    assert st["kind"].startswith("run image") and st["bytes"] < 40e6          # 490 009 steps: 12 MB of registers + 22 MB of memory
    run = api.CairoRun.fibonacci(149000)
    proof = hip_ctx.cairo_prove_run(run, api.ProofOptions(8, 80, 3, 20))
    assert hashlib.sha256(proof).hexdigest() == "3b115b1ab0a2d9e2710d2d8a2f4f4a85938bbe574fe7e5ace903c47040ebaa88"
    assert hip_ctx.last_upload_stats()["bytes"] < 80e6                          # against 1.14 GB of table


def test_rc_builtin_proofs_through_the_device_built_trace(hip_ctx, oracle):
    from test_rc_builtin import PROGRAMS, run_of
    for name in PROGRAMS:
        run = run_of(name)
        options = (4, 3, 3, 1)
        want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
        assert hip_ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want, name
        assert hip_ctx.last_upload_stats()["kind"].startswith("run image")


def test_a_run_rebuilt_from_in_memory_arrays_proves_the_same_bytes(hip_ctx, oracle):
    """sp_cairo_run_from_arrays (cairo-vm's relocated outputs as the reference receives them, lambdaworks limbs) -> sp_cairo_prove_run."""
    from test_rc_builtin import run_of, segments_of
    run = run_of("output_and_rc")
    regs, addrs, values = run.export(api.SP_FE_MONT_LIMBS)
    program_size = run.public_inputs_c.n_public_memory - sum(e - s for t, s, e in segments_of(run) if t == 1)
    again = api.CairoRun.from_arrays(regs, addrs, values, program_size, segments_of(run), api.SP_FE_MONT_LIMBS)
    options = (4, 3, 3, 1)
    want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    assert hip_ctx.cairo_prove_run(again, api.ProofOptions(*options)) == want
    _same(again, hip_ctx)
