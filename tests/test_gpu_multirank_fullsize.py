"""Sharded prover at the sizes BASELINE.json names, against reference-pinned bytes, over two transports: "gloo-staged" (several ranks
sharing the one GPU of the test box, host-staged gloo hooks - what runs on a one-GPU box) and "rccl" (one device per rank, the
library's own RCCL communicator over xGMI: sp_comm_init_rccl - skipped with the reason unless the box has as many GPUs as ranks):

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

from lambdaworks_cairo_prover_amd import api

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
    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only (the 128-byte RCCL id, or the staged hooks)
    try:
        rccl = knobs.get("transport") == "rccl"
        device = rank if rccl else 0
        ctx = api.Context(device=device)
        if rccl:                           # one GPU per rank: ncclAllGather + grouped ncclSend / ncclRecv, blocking and stream-ordered forms
            ctx.init_rccl()
            ctx.comm_selftest(64 << 20)    # rank-stamped 64 MB blocks through every installed primitive
        else:
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
            dev_trace = torch.from_numpy(run.main_trace()).to(f"cuda:{device}")
            torch.cuda.synchronize()
            proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), run.n_rows, run.n_cols, run.public_inputs_c, opt)
        stats = ctx.comm_stats()
        stats.update(ctx.last_proof_info())
        stats["device_bytes"] = ctx.prover_device_bytes()
        stats["link"] = ctx.comm_measure(0)          # what sp_comm_init_rccl measured (zeros under the staged hooks)
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
    try:
        # (a collective that never completes on hardware nobody has run yet must not eat the wall-clock limit of the whole suite)
        got = [q.get(timeout=300 if knobs.get("transport") == "rccl" else 900) for _ in procs]
        for p in procs:
            p.join(timeout=120)
    finally:                                # a rank that hangs (a collective that never completes) must not outlive the test
        for p in procs:
            if p.is_alive():
                p.terminate()               # exactly the processes started above
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
    return {g[0]: g[1:] for g in got}


def _need_devices(world, knobs):
    """The rccl cases need one GPU per rank; on a smaller box they show as skipped, with the reason."""
    if knobs.get("transport") != "rccl":
        return
    import ctypes
    from lambdaworks_cairo_prover_amd import _lib
    n = ctypes.c_int(0)
    _lib.load().sp_device_count(ctypes.byref(n))
    if n.value < world:
        pytest.skip(f"transport rccl / one device per rank: needs {world} GPUs, this box has {n.value}")


def _case_id(v):
    return ",".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v)


CFG4_CASES = [
    # (world, knobs) - options (4, 3, 3, 1): the golden file's
    (4, {"entry": "rows", "want_bytes": True}),                       # all-to-all hook; the host table on every rank: each uploads its role's 9 columns, the trace is all-gathered
    (4, {"entry": "run", "async": True, "shard_interp": 1}),          # block-wise coefficient exchange on the communication stream; the trace built on every rank from the run
    (4, {"entry": "dev", "shard_interp": 0}),                         # every rank interpolates every column
    (4, {"entry": "rows", "alltoall": False, "shard_interp": 1}),     # digest exchange through the all-gather fallback
    (8, {"entry": "rows"}),                                           # ranks 4..7 replicate roles 0..3
    (8, {"entry": "dev", "async": True, "shard_interp": 1}),
    (2, {"entry": "run"}),                                            # two cosets per rank
    # the same over RCCL on distinct devices (benches/criterion_prover_70k.rs:30-57 split as prover.rs:161-185 fans out): every input
    # form, by-column and replicated interpolation, and the mode the MEASURED link rate picks
    (4, {"transport": "rccl", "entry": "rows", "shard_interp": 0, "want_bytes": True}),
    (4, {"transport": "rccl", "entry": "rows", "shard_interp": 1}),
    (4, {"transport": "rccl", "entry": "run", "shard_interp": 0}),
    (4, {"transport": "rccl", "entry": "run", "shard_interp": 1}),
    (4, {"transport": "rccl", "entry": "dev", "shard_interp": 0}),
    (4, {"transport": "rccl", "entry": "dev", "shard_interp": 1}),
    (4, {"transport": "rccl", "entry": "rows"}),
    (8, {"transport": "rccl", "entry": "run"}),                      # ranks 4..7 replicate roles 0..3
    (2, {"transport": "rccl", "entry": "dev"}),                      # two cosets per rank (a two-GPU box runs this one)
]


@pytest.mark.parametrize("world,knobs", CFG4_CASES, ids=_case_id)
def test_config4_split_equals_the_reference_golden_file(world, knobs):
    _need_devices(world, knobs)
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
        if knobs.get("transport") == "rccl":
            link = stats["link"]
            assert link["world"] == world and link["allgather_gbs_per_link"] > 0 and link["alltoall_gbs_per_link"] > 0, link
            if "shard_interp" not in knobs:      # mode 2: the rule on the measured rate (identical on every rank: the minimum over ranks)
                assert stats["interpolation_sharded"] == api.model_shard_interpolation(link["allgather_gbs_per_link"] / api.LINK_MEASURED_MARGIN, min(world, 4), 19)
            else:
                assert stats["interpolation_sharded"] == knobs["shard_interp"]
        else:
            assert stats["interpolation_sharded"] == (1 if knobs.get("shard_interp") == 1 else 0)      # default: the link model (replicated)
        if knobs.get("alltoall", True) and world <= 4:
            assert stats["alltoall_calls"] >= 3
        # 2^21 LDE points at the default knobs: FRI layers 0 .. 5 (>= 2^16 leaves) stay sharded
        assert stats["fri_sharded_layers"] == 6


@pytest.mark.parametrize("world,knobs", [(8, {"entry": "rows"}), (8, {"entry": "dev", "async": True, "shard_interp": 1}),
                                         (8, {"transport": "rccl", "entry": "run"}), (8, {"transport": "rccl", "entry": "rows", "shard_interp": 1})], ids=_case_id)
def test_config3_shape_split_over_eight_ranks(world, knobs):
    _need_devices(world, knobs)
    """2^20 rows x 52 columns, blowup 8, 80 queries, 20-bit grinding on eight ranks (one LDE coset each, ~10 GB per rank):
    the bytes of the one-off CPU-oracle run."""
    results = _run_world(world, "149000", (8, 80, 3, 20), knobs)
    for r in range(world):
        sha, ln, _, stats = results[r]
        assert sha == SHA_CFG3, (r, sha[:600])
        assert stats["groups"] == 8 and stats["composition_path"] == 1 and stats["fri_sharded_layers"] == 8
        assert stats["device_bytes"] < 16e9
