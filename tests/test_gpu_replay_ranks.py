"""tools/replay_ranks.py - a G-rank sharded proof on one GPU, one rank at a time, the collectives replayed from recordings - checked
where the CPU oracle can still follow: configs[4]'s SHAPE (blowup 16, 8 ranks, two cosets each) at 2^14 rows, 4 ranks at blowup 4, and
8 ranks with one coset each at blowup 8; Keccak and Poseidon trees.  Every rank's bytes equal the oracle's and the single-rank device
proof.  The same harness produces the proof of configs[4] at its own size (2^24 rows; profiles/r05_cfg5_replayed_proof.txt)."""
import os
import sys

import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("stream_ordered", [False, True], ids=["blocking", "stream-ordered"])
@pytest.mark.parametrize("fib,blowup,world,poseidon", [(2000, 16, 8, False), (2000, 4, 4, False), (1000, 8, 8, False), (500, 16, 8, True), (300, 2, 8, False)])
def test_replayed_ranks_give_the_oracle_bytes(oracle, hip_ctx, fib, blowup, world, poseidon, stream_ordered):
    from replay_ranks import sharded_proof_by_replay
    run = api.CairoRun.fibonacci(fib)
    options = (blowup, 5, 3, 2)
    opt = api.ProofOptions(*options)
    oracle.set_merkle_backend(1 if poseidon else 0)
    try:
        want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    finally:
        oracle.set_merkle_backend(0)
    with api.Context(device=0) as ctx:
        if poseidon:
            ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
        proofs, stats = sharded_proof_by_replay(api, ctx, lambda c: c.cairo_prove_run(run, opt), world, log=lambda *_: None, stream_ordered=stream_ordered)
        info = ctx.last_proof_info()
    assert sorted(proofs) == list(range(world))
    for r in range(world):
        assert proofs[r] == want, (r, stats)
    assert stats["collectives"] >= 8 and stats["alltoalls"] >= (3 if world <= blowup else 0) and info["groups"] == min(world, blowup)
    assert stats["partial_runs"] == (stats["collectives"] + 1) * world
    assert api.cairo_verify(want, run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON if poseidon else api.SP_MERKLE_KECCAK256)


SHA_CFG3 = "3b115b1ab0a2d9e2710d2d8a2f4f4a85938bbe574fe7e5ace903c47040ebaa88"     # tests/golden/README_config3.md (a one-off CPU-oracle run)


@pytest.mark.parametrize("world,stream_ordered,shard_interp,entry", [(2, False, 2, "run"), (4, True, 2, "run"), (4, False, 1, "run"), (2, True, 1, "run"),
                                                                     (4, False, 2, "rows"), (2, True, 1, "rows")])
def test_config3_at_full_size_on_two_and_four_ranks(world, stream_ordered, shard_interp, entry):
    """BASELINE configs[2] (2^20 rows, blowup 8, 80 queries, 20-bit grinding) split over 2 and 4 ranks - four and two LDE cosets per rank:
    the world sizes the driver's scaling run takes between 1 and 8, which the concurrent shared-GPU cases (test_gpu_multirank_fullsize.py)
    cover at 8 only.  Replayed ranks, blocking and stream-ordered hooks, replicated and by-column interpolation, from the run (the trace
    built on every rank) and from the reference's row-major host table (every rank uploads its share of the columns, the trace is
    all-gathered): the pinned digest."""
    import hashlib
    from replay_ranks import sharded_proof_by_replay
    run = api.CairoRun.fibonacci(149000)
    opt = api.ProofOptions(8, 80, 3, 20)
    trace = run.main_trace() if entry == "rows" else None
    prove = (lambda c: c.cairo_prove(trace, run.public_inputs_c, opt)) if entry == "rows" else (lambda c: c.cairo_prove_run(run, opt))
    with api.Context(device=0) as ctx:
        ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, shard_interp)
        proofs, stats = sharded_proof_by_replay(api, ctx, prove, world, log=lambda *_: None, stream_ordered=stream_ordered)
        info = ctx.last_proof_info()
    assert sorted(proofs) == list(range(world))
    for r in range(world):
        assert hashlib.sha256(proofs[r]).hexdigest() == SHA_CFG3, (r, stats)
    assert info["groups"] == world and info["interpolation_sharded"] == (1 if shard_interp == 1 else 0)
