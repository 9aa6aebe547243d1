"""Random hint-free Cairo programs (tests/cairo_asm.py: random_program - immediates near 0 and near p, sums and products through ap-based
operands, `ap += k` holes, taken and not-taken jumps, calls): the front-end's run and main trace (reference src/cairo/runner/run.rs:242-263,
src/cairo/execution_trace.rs:57-87) satisfy the Cairo AIR - the CPU oracle proves them and both verifiers accept."""
import pytest

import cairo_asm as A
from lambdaworks_cairo_prover_amd import api

OPTIONS = (4, 3, 3, 1)


@pytest.mark.parametrize("seed", range(12))
def test_random_program_runs_prove_and_verify_on_the_cpu(oracle, seed):
    words, entry = A.random_program(seed, length=20 + 3 * seed)
    run = api.CairoRun.from_program(words, entry_pc=entry)
    assert run.n_cols == 34 and run.num_steps >= 20
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, OPTIONS)
    assert oracle.cairo_verify(proof, run.public_inputs_c, OPTIONS)
    assert api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(*OPTIONS))


def test_the_generator_reaches_holes_and_both_jump_directions():
    holes = jumps_taken = jumps_not = 0
    for seed in range(12):
        words, entry = A.random_program(seed, length=20 + 3 * seed)
        run = api.CairoRun.from_program(words, entry_pc=entry)
        t = run.main_trace()
        steps = run.num_steps
        jnz = t[:steps, 9, 31] == 1                       # f_pc_jnz (air.rs:29-46: flag column 9)
        taken = jnz & (t[:steps, 24:25, :].reshape(steps, 32).any(axis=1))     # dst != 0
        jumps_taken += int(taken.sum())
        jumps_not += int((jnz & ~taken).sum())
        holes += int(run.n_rows > (1 << (steps - 1).bit_length()))
    assert jumps_taken > 0 and jumps_not > 0 and holes > 0
