"""Sharded prover at the sizes BASELINE.json names, against reference-pinned bytes (several ranks sharing the one GPU of the
test box, host-staged gloo hooks):

* configs[3] - the 70k program of benches/criterion_prover_70k.rs (2^19 rows, blowup 4) split over 4 ranks (one coset each) and
  over 8 (replicas beyond the blowup factor): every rank's bytes equal tests/golden/fibonacci_70000.proof, the file the
  reference itself produced - with the digest all-to-all, with the block-wise stream-ordered coefficient exchange, with
  column-sharded and with replicated interpolation, and from each of the input forms (the reference's row-major host table,
  the run's own columns, a table already in HBM);
* configs[2] - 2^20 rows, blowup 8, 80 queries, 20-bit grinding - over 8 ranks (one coset each): the sha256 pinned by the
  one-off CPU-oracle run of tests/golden/README_config3.md.

What the toy-sized cases of test_gpu_multirank.py cannot reach: block counts and overlapping column ranges of the coefficient
exchange at 34 / 18 columns of 2^19 and 2^20 rows, digest blocks of 2^18 ... 2^20 leaves, sharded FRI layers down to 2^16 leaves
at the default knobs, the row-sharded trace check on real slices."""
import hashlib
import os
import socket

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHA_70K = "da962bd4513d991c39a0e0cc11cc76d25b9ec405cebcdaaf1449184d4b54cd6b"      # tests/golden/fibonacci_70000.proof, proof section
SHA_CFG3 = "3b115b1ab0a2d9e2710d2d8a2f4f4a85938bbe574fe7e5ace903c47040ebaa88"     # tests/golden/README_config3.md


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_of(case):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from lambdaworks_cairo_prover_amd import api
    if case == "golden70k":     # the program inside the golden file's own public inputs (22 words, no final assert)
        from test_gpu_prover import program_words_from_proof_file
        _, words = program_words_from_proof_file(os.path.join(GOLDEN, "fibonacci_70000.proof"))
        return api.CairoRun.from_program(words)
    return api.CairoRun.fibonacci(int(case))


def _worker(rank, world, port, case, options, knobs, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = api.Context(device=0)
        ctx.set_collective(world, rank, api.StagedAllGather(), alltoall=knobs.get("alltoall", True))
        if knobs.get("async"):
            ctx.set_collective_async(api.StagedAsyncAllGather())
        if "shard_interp" in knobs:
            ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, knobs["shard_interp"])
        run = _run_of(case)
        opt = api.ProofOptions(*options)
        entry = knobs.get("entry", "rows")
        if entry == "rows":            # the reference's row-major TraceTable in pageable host memory (sp_cairo_prove)
            proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, opt)
        elif entry == "run":           # the front-end's own column-major store (sp_cairo_prove_run)
            proof = ctx.cairo_prove_run(run, opt)
        else:                          # already in HBM (sp_cairo_prove_dev)
            import torch
            dev_trace = torch.from_numpy(run.main_trace()).to("cuda:0")
            torch.cuda.synchronize()
            proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), run.n_rows, run.n_cols, run.public_inputs_c, opt)
        stats = ctx.comm_stats()
        stats.update(ctx.last_proof_info())
        stats["device_bytes"] = ctx.prover_device_bytes()
        q.put((rank, hashlib.sha256(proof).hexdigest(), len(proof), proof if knobs.get("want_bytes") and rank == 0 else None, stats))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0, None, None))
    finally:
        dist.destroy_process_group()


def _run_world(world, case, options, knobs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, options, knobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    return {g[0]: g[1:] for g in got}


CFG4_CASES = [
    # (world, knobs) - options (4, 3, 3, 1): the golden file's
    (4, {"entry": "rows", "want_bytes": True}),                       # all-to-all hook; the host table on every rank: each uploads its role's 9 columns, the trace is all-gathered
    (4, {"entry": "run", "async": True, "shard_interp": 1}),          # block-wise coefficient exchange on the communication stream; the trace built on every rank from the run
    (4, {"entry": "dev", "shard_interp": 0}),                         # every rank interpolates every column
    (4, {"entry": "rows", "alltoall": False, "shard_interp": 1}),     # digest exchange through the all-gather fallback
    (8, {"entry": "rows"}),                                           # ranks 4..7 replicate roles 0..3
    (8, {"entry": "dev", "async": True, "shard_interp": 1}),
    (2, {"entry": "run"}),                                            # two cosets per rank
]


@pytest.mark.parametrize("world,knobs", CFG4_CASES)
def test_config4_split_equals_the_reference_golden_file(world, knobs):
    golden = open(os.path.join(GOLDEN, "fibonacci_70000.proof"), "rb").read()
    plen = int.from_bytes(golden[:8], "big")
    want = golden[8:8 + plen]
    assert hashlib.sha256(want).hexdigest() == SHA_70K
    results = _run_world(world, "golden70k", (4, 3, 3, 1), knobs)
    for r in range(world):
        sha, ln, proof, stats = results[r]
        assert sha == SHA_70K, (r, sha[:600])
        assert ln == len(want)
        if proof is not None:
            assert proof == want
        assert stats["world"] == world and stats["groups"] == min(world, 4) and stats["composition_path"] == 1
        assert stats["interpolation_sharded"] == (1 if knobs.get("shard_interp") == 1 else 0)      # default: the link model (replicated)
        if knobs.get("alltoall", True) and world <= 4:
            assert stats["alltoall_calls"] >= 3
        # 2^21 LDE points at the default knobs: FRI layers 0 .. 5 (>= 2^16 leaves) stay sharded
        assert stats["fri_sharded_layers"] == 6


@pytest.mark.parametrize("world,knobs", [(8, {"entry": "rows"}), (8, {"entry": "dev", "async": True, "shard_interp": 1})])
def test_config3_shape_split_over_eight_ranks(world, knobs):
    """2^20 rows x 52 columns, blowup 8, 80 queries, 20-bit grinding on eight ranks (one LDE coset each, ~10 GB per rank):
    the bytes of the one-off CPU-oracle run."""
    results = _run_world(world, "149000", (8, 80, 3, 20), knobs)
    for r in range(world):
        sha, ln, _, stats = results[r]
        assert sha == SHA_CFG3, (r, sha[:600])
        assert stats["groups"] == 8 and stats["composition_path"] == 1 and stats["fri_sharded_layers"] == 8
        assert stats["device_bytes"] < 16e9
