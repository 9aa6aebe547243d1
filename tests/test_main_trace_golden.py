"""Host front-end (mini Cairo VM + main-trace builder) against the expected tables of the reference's own unit tests
(src/cairo/execution_trace.rs:661-1161; fixture made by tests/golden/make_main_trace_vectors.py), and the dump parsers
against the reference's tests/data files (register_states.rs:155-186, cairo_mem.rs:121-135)."""
import json
import os

from lambdaworks_cairo_prover_amd import api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = api.P

# bytecode of cairo_programs/cairo0/simple_program.cairo and call_func.cairo as read off the instruction / immediate
# columns of the expected tables (pc column 19, instruction column 23, op1 column 26 where op1_addr = pc + 1)
SIMPLE = [0x480680017FFF8000, 3, 0x400680017FFF7FFF, 3, 0x208B7FFF7FFF7FFE]
CALL_FUNC = [0x484A7FFD7FFC8000, 0x208B7FFF7FFF7FFE,            # mul: [ap] = [fp-3] * [fp-4]; ap++ ; ret
             0x480680017FFF8000, 2, 0x480680017FFF8000, 3,      # main: [ap] = 2; ap++ ; [ap] = 3; ap++
             0x1104800180018000, P - 6,                         # call rel -6
             0x400680017FFF7FFF, 6, 0x208B7FFF7FFF7FFE]         # assert [ap-1] = 6 ; ret


def _tables():
    with open(os.path.join(GOLDEN, "main_trace_tables.json")) as f:
        return json.load(f)


def _check(words, entry_pc, table):
    cols = [[int(v, 16) for v in c] for c in table["columns"]]
    steps = len(cols[0])
    run = api.CairoRun.from_program(words, max_steps=64, entry_pc=entry_pc)
    assert run.num_steps == steps
    assert run.n_cols == 34
    got = run.main_trace()
    for j in range(34):
        col = api.bytes_to_felts(got[:steps, j])
        assert col == cols[j], f"column {j}: {col} != {cols[j]}"


def test_program_words_match_table_columns():
    """The hand-transcribed bytecode agrees with the instruction/immediate columns of the fixture."""
    for words, name in ((SIMPLE, "simple_program"), (CALL_FUNC, "call_func_program")):
        cols = [[int(v, 16) for v in c] for c in _tables()[name]["columns"]]
        for pc, inst, op1_addr, op1 in zip(cols[19], cols[23], cols[22], cols[26]):
            assert words[pc - 1] == inst
            if op1_addr == pc + 1:
                assert words[pc] == op1


def test_main_trace_simple_program(hip_lib):
    _check(SIMPLE, 1, _tables()["simple_program"])


def test_main_trace_call_func_program(hip_lib):
    _check(CALL_FUNC, 3, _tables()["call_func_program"])


def test_dump_parsers_mul_program(hip_lib):
    trace = open(os.path.join(GOLDEN, "mul_trace.out"), "rb").read()
    memory = open(os.path.join(GOLDEN, "mul_mem.out"), "rb").read()
    # register_states.rs:155-186: rows (ap, fp, pc) = (8, 8, 1), (9, 8, 3), (9, 8, 5)
    assert len(trace) == 3 * 24
    rows = [tuple(int.from_bytes(trace[24 * i + 8 * k:24 * i + 8 * k + 8], "little") for k in range(3)) for i in range(3)]
    assert rows == [(8, 8, 1), (9, 8, 3), (9, 8, 5)]
    # cairo_mem.rs:121-135: addresses are 1..len contiguous
    addrs = sorted(int.from_bytes(memory[40 * i:40 * i + 8], "little") for i in range(len(memory) // 40))
    assert addrs == list(range(1, len(addrs) + 1))
    run = api.CairoRun.from_dumps(trace, memory, program_size=5)
    assert run.num_steps == 3
    got = run.main_trace()
    assert api.bytes_to_felts(got[:3, 19]) == [1, 3, 5]       # pc
    assert api.bytes_to_felts(got[:3, 17]) == [8, 9, 9]       # ap
    assert api.bytes_to_felts(got[:3, 18]) == [8, 8, 8]       # fp
    # re-running the dumped program words through the VM reproduces the dumped run
    words = {int.from_bytes(memory[40 * i:40 * i + 8], "little"): int.from_bytes(memory[40 * i + 8:40 * i + 40], "little")
             for i in range(len(memory) // 40)}
    vm = api.CairoRun.from_program([words[a] for a in range(1, 6)], max_steps=64).main_trace()
    assert (vm == got).all()


def test_dump_parsers_reject_truncated_files(hip_lib):
    trace = open(os.path.join(GOLDEN, "mul_trace.out"), "rb").read()
    memory = open(os.path.join(GOLDEN, "mul_mem.out"), "rb").read()
    import pytest
    with pytest.raises(api.SpError):   # IncorrectNumberOfBytes (register_states.rs:137-151)
        api.CairoRun.from_dumps(trace[:-1], memory, program_size=5)
    with pytest.raises(api.SpError):   # IncorrectNumberOfBytes (cairo_mem.rs:105-118)
        api.CairoRun.from_dumps(trace, memory[:-1], program_size=5)


def test_run_column_store_matches_the_row_major_trace():
    """sp_cairo_run_columns: the run's own store is the column-major device-layout form of sp_cairo_run_main_trace."""
    import ctypes
    import numpy as np
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(300)
    addr, n, c, _pinned = run.columns()
    assert (n, c) == (run.n_rows, run.n_cols)
    raw = np.frombuffer(ctypes.string_at(addr, n * c * 32), dtype=np.uint8).reshape(c, n, 32)
    be = api.fe_from_device(raw.reshape(-1, 32)).reshape(c, n, 32)
    assert np.array_equal(be.transpose(1, 0, 2), run.main_trace())
    # the lambdaworks limb encoding of the same table
    lw = run.main_trace(api.SP_FE_MONT_LIMBS)
    assert np.array_equal(api.fe_to_device(lw.reshape(-1, 32), api.SP_FE_MONT_LIMBS).reshape(n, c, 32), raw.transpose(1, 0, 2))


def test_run_builds_its_host_table_on_demand_only(hip_lib):
    """A run validates its trace and fixes its shape at creation (the shape pass of build_main_trace) but writes the n x cols host
    table only when something asks for it - sp_cairo_prove_run builds the trace on the device from the register states and the
    memory (sp_cairo_run_timings reports the split)."""
    run = api.CairoRun.fibonacci(500)
    t = run.timings()
    assert set(t) == {"vm_ms", "trace_shape_ms", "device_image_ms", "host_table_ms"}
    assert t["host_table_ms"] == 0.0 and t["vm_ms"] > 0.0 and run.n_rows == 4096 and run.n_cols == 34
    table = run.main_trace()
    assert table.shape == (4096, 34, 32) and run.timings()["host_table_ms"] > 0.0
    # a program that fails validation fails at run creation, as it did when the table was built eagerly
    import pytest
    with pytest.raises(api.SpError):
        api.CairoRun.from_program([0x480680017FFF8000, 3, 0x400680017FFF7FFF, 4, 0x208B7FFF7FFF7FFE], max_steps=64)   # assert 3 == 4
