"""Valid programs that use the range_check (and output) builtins - the layout of the reference's `rc_program`, `lt_comparison`
and `signed_div_rem` tests (tests/integration_tests.rs:151-172; 43 main columns, 50 constraints, src/cairo/air.rs:623-629,
1141-1160; builder src/cairo/execution_trace.rs:358-379, 604-624).  CPU side: the product's VM + main-trace builder produce
traces on which all 50 constraints vanish, the oracle proves them and both verifiers accept."""
import ctypes

import numpy as np
import pytest

import cairo_asm as A
import oracle_lib
from lambdaworks_cairo_prover_amd import api

# air.rs:605-616 + the range-check builtin constraint (no exemption)
EXEMPTIONS = [0] * 16 + [0] + [0, 0, 0] + [1, 1, 1, 1, 0, 0] + [0] * 5 + [0, 0, 0, 1] * 3 + [0, 0, 1] + [0, 0, 0] + [0]

PROGRAMS = {
    "rc_program": (A.rc_program, {}),
    "rc_loop_20": (lambda: A.rc_loop_program(20), {}),
    "rc_loop_300": (lambda: A.rc_loop_program(300, start=2**127 - 1500, step=1), {}),
    "output_and_rc": (A.output_rc_program, {"output": True}),
}


def run_of(name):
    make, kw = PROGRAMS[name]
    words, entry = make()
    return api.CairoRun.from_program_builtins(words, entry_pc=entry, **kw)


def segments_of(run):
    pi = run.public_inputs_c
    types = list(ctypes.string_at(pi.segment_types, pi.n_segments))
    ranges = list((ctypes.c_uint64 * (2 * pi.n_segments)).from_address(pi.segment_ranges))
    return [(types[i], ranges[2 * i], ranges[2 * i + 1]) for i in range(pi.n_segments)]


def test_vm_lays_out_the_builtin_segments(hip_lib):
    run = run_of("rc_program")
    assert run.n_cols == 43 and run.num_steps == 12
    (kind, start, end), = segments_of(run)
    assert kind == 0 and end - start == 2          # two range-checked values
    main = run.main_trace()
    vals = [int.from_bytes(main[i, 42].tobytes(), "big") for i in range(3)]
    assert vals == [5, 2, 0]                        # rc_value column: the segment's cells, then zero padding
    assert [int.from_bytes(main[0, 34 + k].tobytes(), "big") for k in range(8)] == [5, 0, 0, 0, 0, 0, 0, 0]
    run = run_of("output_and_rc")
    segs = segments_of(run)
    assert [s[0] for s in segs] == [1, 0] and segs[0][2] == segs[1][1]      # output segment, then range_check right behind it
    pm = dict(run.public_memory())
    assert pm[segs[0][1]] == 7 and pm[segs[0][1] + 1] == 2**100 + 9          # the output cells are public memory (air.rs:200-206)
    big = run_of("rc_loop_300").main_trace()
    v = int.from_bytes(big[7, 42].tobytes(), "big")
    assert v == 2**127 - 1500 + 7
    assert sum(int.from_bytes(big[7, 34 + k].tobytes(), "big") << (16 * k) for k in range(8)) == v


def test_range_check_rejects_values_beyond_128_bits(hip_lib):
    words, entry = A.rc_loop_program(3, start=2**128 - 1, step=1)   # the second value is 2^128
    with pytest.raises(api.SpError):
        api.CairoRun.from_program_builtins(words, entry_pc=entry)
    words, entry = A.rc_loop_program(3, start=-1, step=0)
    with pytest.raises(api.SpError):
        api.CairoRun.from_program_builtins(words, entry_pc=entry)


@pytest.mark.parametrize("name", ["rc_program", "rc_loop_20", "output_and_rc"])
def test_all_50_constraints_vanish(oracle, hip_lib, name):
    run = run_of(name)
    rap = (2**250 + 12345, 2**249 + 99, 2**200 + 1)
    main = run.main_trace()
    aux = oracle_lib.cairo_aux_trace(main, run.public_inputs_c, rap)
    full = np.concatenate([main, aux], axis=1)
    n = full.shape[0]
    zero = bytes(32)
    for i in range(n):
        ev = oracle_lib.cairo_transition(np.stack([full[i], full[(i + 1) % n]]), rap, has_rc_builtin=True)
        for c in range(50):
            if i >= n - EXEMPTIONS[c]:
                continue
            assert ev[c].tobytes() == zero, f"constraint {c} does not vanish at row {i}"


@pytest.mark.parametrize("name,options", [("rc_program", (4, 3, 3, 1)), ("rc_loop_20", (4, 5, 3, 2)), ("output_and_rc", (8, 3, 3, 1)),
                                          ("rc_loop_300", (2, 4, 3, 1))])
def test_oracle_proves_and_both_verifiers_accept(oracle, hip_lib, name, options):
    run = run_of(name)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    assert oracle.cairo_verify(proof, run.public_inputs_c, options)
    assert api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(*options))
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not api.cairo_verify(bytes(bad), run.public_inputs_c, api.ProofOptions(*options))


def test_runs_round_trip_through_the_in_memory_arrays(hip_lib):
    """sp_cairo_run_export / sp_cairo_run_from_arrays - the form in which cairo-vm hands its relocated register states, memory and
    builtin segments to the reference (run.rs:64-263): a run rebuilt from its own arrays has the same shape, public inputs and main
    trace, in both encodings, with and without builtin segments; malformed inputs are rejected."""
    import numpy as np
    for name in ("rc_program", "output_and_rc", "rc_loop_20"):
        run = run_of(name)
        for enc in (api.SP_FE_CANON_BE, api.SP_FE_MONT_LIMBS):
            regs, addrs, values = run.export(enc)
            assert regs.shape == (run.num_steps, 3) and len(addrs) == len(values) and (np.diff(addrs.astype(np.int64)) > 0).all()
            program_size = run.public_inputs_c.n_public_memory - sum(e - s for t, s, e in segments_of(run) if t == 1)
            again = api.CairoRun.from_arrays(regs, addrs, values, program_size, segments_of(run), enc)
            assert (again.n_rows, again.n_cols, again.num_steps) == (run.n_rows, run.n_cols, run.num_steps)
            assert again.public_memory() == run.public_memory() and segments_of(again) == segments_of(run)
            assert np.array_equal(again.main_trace(), run.main_trace())
    plain = api.CairoRun.fibonacci(50)
    regs, addrs, values = plain.export()
    again = api.CairoRun.from_arrays(regs, addrs, values, 22)
    assert np.array_equal(again.main_trace(), plain.main_trace())
    with pytest.raises(api.SpError):        # a cell the trace reads is missing
        api.CairoRun.from_arrays(regs, addrs[:-30], values[:-30], 22)
    with pytest.raises(api.SpError):        # a segment that ends before it starts
        api.CairoRun.from_arrays(regs, addrs, values, 22, [(0, 100, 90)])
