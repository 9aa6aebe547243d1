"""Sharded prover: world_size 2, 4 and 8 ranks produce the SAME proof bytes as the single-rank prover and
the CPU oracle - for valid and for constraint-violating traces, with the digest exchange as an all-to-all or through its
all-gather fallback, with sharded and with replicated FRI layers, with column-sharded and with replicated interpolation, and
with more ranks than cosets.  On the one-GPU test box the ranks share device 0 and exchange through the host-staged gloo hooks
(`test_rccl_on_distinct_devices` runs the library's own RCCL communicator when the box has two GPUs or more)."""
import os
import random
import socket

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(case):
    """(main trace, public inputs for the product, options) of a named case, rebuilt identically in every process."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from lambdaworks_cairo_prover_amd import api
    kind = case["kind"]
    if kind == "fib":
        run = api.CairoRun.fibonacci(case["fib"])
        return run.main_trace(), run.public_inputs_c, run
    if kind == "fib_flip":     # a valid trace with ONE cell changed: only the rank that checks that row sees the violation
        run = api.CairoRun.fibonacci(case["fib"])
        trace = run.main_trace().copy()
        trace[case["row"], case["col"], 31] ^= 1
        return trace, run.public_inputs_c, run
    if kind == "prog":         # a random hint-free program (cairo_asm.random_program), optionally with one cell replaced
        import cairo_asm as A
        import numpy as np
        words, entry = A.random_program(case["seed"], length=case.get("length", 40))
        run = api.CairoRun.from_program(words, entry_pc=entry)
        trace = run.main_trace().copy()
        if "cell" in case:
            r, c, v = case["cell"]
            trace[r % trace.shape[0], c] = np.frombuffer(int(v).to_bytes(32, "big"), dtype=np.uint8)
        return trace, run.public_inputs_c, run
    import oracle_lib as oracle
    from test_gpu_random_traces import random_trace
    rng = random.Random(case["seed"])
    cols = 43 if case.get("rc") else 34
    n = case["n"]
    trace = random_trace(rng, n, cols)
    pm = [(a, rng.randrange(api.P)) for a in range(1, 6)]
    segments = [(0, 1000, 1010)] if case.get("rc") else []
    pub, keep = oracle.make_public_inputs(rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100), rng.randrange(1, 100),
                                          rng.randrange(1, 100), 5, 65000, pm, n - 7, segments)
    return trace, pub, keep


def _worker(rank, world, port, case, options, knobs, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if knobs.get("rows_window"):
        os.environ["SP_UPLOAD_MIN_MB"] = "0"      # (read once per process by the library)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        trace, pub, keep = _inputs(case)
        ctx = api.Context(device=0)
        ctx.set_collective(world, rank, api.StagedAllGather(), alltoall=knobs.get("alltoall", True))
        if knobs.get("async"):   # stream-ordered all-gather: the coefficient exchange of a segment goes in column blocks beside the transforms
            ctx.set_collective_async(api.StagedAsyncAllGather(alltoall=knobs.get("async_a2a", True)))
        ctx.comm_selftest(4096)  # rank-stamped blocks through every installed primitive (blocking and stream-ordered)
        link = ctx.comm_measure(1 << 16) if knobs.get("measure") else None   # sp_comm_measure through a transport that really exchanges
        if "measure_fault" in knobs:     # one rank cannot prepare its payload buffers (the library's fault injection): what do the OTHERS do?
            os.environ["SP_COMM_MEASURE_FAULT_RANK"] = str(knobs["measure_fault"])
            try:
                ctx.comm_measure(1 << 16)
                raised = None
            except api.SpError as e:
                raised = (e.code, str(e))
            os.environ.pop("SP_COMM_MEASURE_FAULT_RANK")
            link = dict(ctx.comm_measure(0), raised=raised)
        if "fri_min_log" in knobs:
            ctx.set_option(api.SP_OPT_FRI_SHARD_MIN_LOG, knobs["fri_min_log"])
        if "shard_interp" in knobs:
            ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, knobs["shard_interp"])
        if knobs.get("poseidon"):
            ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
        if knobs.get("prewarm"):   # every rank pre-warms (its small proofs are sharded proofs over the same transport)
            ctx.prewarm(trace.shape[0], trace.shape[1], 18, trace.shape[1] == 43, api.ProofOptions(*options))
        proof = ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
        proof2 = ctx.cairo_prove(trace, pub, api.ProofOptions(*options))  # buffer reuse path
        stats = ctx.comm_stats()
        stats["composition_path"] = ctx.last_proof_info()["composition_path"]
        stats["interpolation_sharded"] = ctx.last_proof_info()["interpolation_sharded"]
        stats["upload_kind"] = ctx.last_upload_stats()["kind"]
        stats["link"] = link
        stats["comm_ms"] = ctx.comm_time_ms()
        q.put((rank, proof if proof == proof2 else b"MISMATCH-ON-REUSE", stats))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, ("fail: " + traceback.format_exc()).encode(), None))
    finally:
        dist.destroy_process_group()


def _run_world(world, case, options, knobs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, options, knobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return {r: (proof, stats) for r, proof, stats in got}


FIB = lambda i: {"kind": "fib", "fib": i}          # noqa: E731
RND = lambda n, seed, rc=False: {"kind": "random", "n": n, "seed": seed, "rc": rc}   # noqa: E731

CASES = [
    # (world, case, options, knobs)
    (2, FIB(100), (4, 3, 3, 1), {"fri_min_log": 5, "shard_interp": 1}),         # interpolation by column + coefficient all-gather
    (4, FIB(100), (4, 5, 3, 2), {"fri_min_log": 6}),                             # default: the link model decides (replicated on 46 GB/s links)
    (4, FIB(200), (8, 4, 3, 1), {"fri_min_log": 5, "alltoall": False, "shard_interp": 1}),   # digest exchange through the all-gather fallback
    (2, FIB(60), (2, 3, 3, 1), {"fri_min_log": 4, "shard_interp": 0}),           # replicated interpolation
    (8, FIB(100), (8, 4, 3, 1), {"fri_min_log": 8, "shard_interp": 1}),          # one coset per rank
    (8, FIB(60), (16, 3, 3, 1), {"fri_min_log": 8}),                             # two cosets per rank (configs[4]'s shape in small)
    (8, FIB(100), (4, 3, 3, 1), {"fri_min_log": 7}),                             # more ranks than cosets: ranks 4..7 replicate roles 0..3
    (4, FIB(140), (4, 6, 3, 1), {}),                                             # default knobs: the whole FRI replicated at this size
    (2, RND(128, 11), (4, 3, 3, 1), {"fri_min_log": 5}),                         # constraint-violating traces (deg H >= 2n)
    (4, RND(256, 12), (8, 4, 3, 2), {"fri_min_log": 6}),
    (4, RND(128, 13, rc=True), (4, 3, 3, 1), {"fri_min_log": 5, "alltoall": False}),
    (8, RND(256, 14), (8, 3, 3, 1), {}),
    # stream-ordered all-gather hook: 17 (9) columns per role in four blocks, 9 (5) in three, replicas beyond the blowup factor
    (2, FIB(100), (4, 3, 3, 1), {"fri_min_log": 5, "async": True, "shard_interp": 1}),
    (4, FIB(200), (8, 4, 3, 1), {"fri_min_log": 5, "async": True, "shard_interp": 1}),
    (8, FIB(100), (4, 3, 3, 1), {"fri_min_log": 7, "async": True, "shard_interp": 1}),
    (4, RND(256, 15, rc=True), (8, 4, 3, 2), {"fri_min_log": 6, "async": True, "shard_interp": 1}),
    # stream-ordered transport: every exchange of a commitment sits on the compute stream and the FRI commit phase runs through its
    # sharded layers without a host round trip (device-side transcript step behind the top tree); without the stream-ordered
    # all-to-all the digest exchange falls back to the stream-ordered all-gather
    (4, FIB(200), (8, 4, 3, 1), {"fri_min_log": 4, "async": True}),
    (8, FIB(100), (8, 4, 3, 1), {"fri_min_log": 5, "async": True, "async_a2a": False}),
    (2, RND(128, 19), (4, 3, 3, 1), {"fri_min_log": 4, "async": True}),
    # sp_prewarm on every rank before the proof (blocking and stream-ordered transport)
    (4, FIB(200), (8, 4, 3, 1), {"fri_min_log": 5, "prewarm": True}),
    (2, FIB(100), (4, 3, 3, 1), {"fri_min_log": 5, "prewarm": True, "async": True, "shard_interp": 1}),
    # the exact trace check of round 2 is split by rows over the ranks (n >= 256 world): a violation in the last rank's slice only
    (2, {"kind": "fib_flip", "fib": 100, "row": 700, "col": 24}, (4, 3, 3, 1), {"fri_min_log": 5}),
    (4, {"kind": "fib_flip", "fib": 100, "row": 3, "col": 24}, (4, 3, 3, 1), {"fri_min_log": 5}),
    (2, RND(1024, 16), (4, 3, 3, 1), {"fri_min_log": 6}),
    # Poseidon commitments (the optional backend of configs[4]): subtrees, digest all-to-all and top tree over field-element digests
    (2, FIB(100), (4, 3, 3, 1), {"fri_min_log": 5, "poseidon": True}),
    (8, FIB(60), (16, 3, 3, 1), {"fri_min_log": 8, "poseidon": True}),
    (4, RND(256, 17), (8, 4, 3, 2), {"fri_min_log": 6, "poseidon": True, "async": True, "shard_interp": 1}),
    # the row-major host table with several ranks: every rank uploads the columns of its role only, the trace columns are all-gathered
    # (SP_UPLOAD_MIN_MB=0 sends these small tables down the path the big ones take)
    (2, FIB(100), (4, 3, 3, 1), {"fri_min_log": 5, "rows_window": True}),
    (4, FIB(200), (8, 4, 3, 1), {"fri_min_log": 5, "rows_window": True, "shard_interp": 1}),
    (8, FIB(100), (4, 3, 3, 1), {"fri_min_log": 7, "rows_window": True}),
    (4, RND(256, 18, rc=True), (8, 4, 3, 2), {"fri_min_log": 6, "rows_window": True, "async": True, "shard_interp": 1}),
]


@pytest.mark.parametrize("world,case,options,knobs", CASES)
def test_sharded_proof_bytes_identical(world, case, options, knobs, oracle, hip_ctx):
    from lambdaworks_cairo_prover_amd import api
    trace, pub, keep = _inputs(case)
    backend = 1 if knobs.get("poseidon") else 0
    oracle.set_merkle_backend(backend)
    hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, backend)
    try:
        want = oracle.cairo_prove(trace, pub, options)
        single = hip_ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
    finally:
        oracle.set_merkle_backend(0)
        hip_ctx.set_option(api.SP_OPT_MERKLE_BACKEND, 0)
    assert single == want
    results = _run_world(world, case, options, knobs)
    for r in range(world):
        proof, stats = results[r]
        assert proof == want, (r, proof[:300])
        assert stats["world"] == world and stats["allgather_calls"] > 0
        # valid traces take the 2n-point composition (also with one coset per rank: cosets 0 and b/2 come from two ranks),
        # constraint-violating ones the whole domain with the general split
        assert stats["composition_path"] == (1 if case["kind"] == "fib" else 3)     # (one flipped cell: deg H >= 2n, like a random trace)
        if knobs.get("alltoall", True) and world <= options[0]:
            assert stats["alltoall_calls"] >= 3          # main, aux and composition commitments at least
        assert stats["interpolation_sharded"] == (1 if knobs.get("shard_interp") == 1 else 0)
        if knobs.get("async") and knobs.get("shard_interp") == 1:
            assert stats["allgather_calls"] >= 2 * (3 + 3)   # two proofs, each segment's coefficients in three or four blocks
        if knobs.get("rows_window"):
            assert stats["upload_kind"].startswith("row-major")   # (the one-copy path reports "single copy")


PROG = lambda seed, cell=None, length=40: dict({"kind": "prog", "seed": seed, "length": length}, **({"cell": cell} if cell else {}))   # noqa: E731


@pytest.mark.parametrize("world,case,options,knobs", [
    (2, PROG(301), (4, 3, 3, 1), {"fri_min_log": 5}),                                              # a valid random program
    (4, PROG(302, length=70), (8, 4, 3, 2), {"fri_min_log": 5, "async": True}),
    (4, PROG(303, cell=(9, 20, (1 << 200) + 12345)), (4, 3, 3, 1), {"fri_min_log": 5, "async": True}),   # an address beyond 2^64: four-limb sort on every rank
    (8, PROG(304, cell=(3, 28, (1 << 16) + 3)), (8, 3, 3, 1), {"fri_min_log": 6}),                  # an offset beyond 16 bits: its low 16 bits are sorted
    (4, PROG(305, cell=(17, 21, 1 << 64)), (4, 3, 3, 1), {"fri_min_log": 5, "rows_window": True}),  # ... through the windowed row-major upload
])
def test_sharded_random_programs(world, case, options, knobs, oracle, hip_ctx):
    """Random programs - valid, and with one address / offset cell outside anything a VM writes - through the sharded prover: every
    rank gives the oracle's bytes (the auxiliary trace is built whole on every rank, wide-address fallback included)."""
    trace, pub, keep = _inputs(case)
    want = oracle.cairo_prove(trace, pub, options)
    results = _run_world(world, case, options, knobs)
    for r in range(world):
        proof, stats = results[r]
        assert proof == want, (r, proof[:300])
        assert stats["composition_path"] == (1 if "cell" not in case else stats["composition_path"])
        assert "cell" not in case or stats["composition_path"] != 1


def _rccl_worker(rank, world, port, fib_index, options, q):
    import sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    try:
        run = api.CairoRun.fibonacci(fib_index)
        ctx = api.Context(device=rank)
        ctx.set_option(api.SP_OPT_FRI_SHARD_MIN_LOG, 8)
        ctx.init_rccl()                       # ncclAllGather / grouped ncclSend+ncclRecv on the context stream
        ctx.comm_selftest(1 << 20)
        proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
        q.put((rank, proof, ctx.comm_stats()))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, ("fail: " + traceback.format_exc()).encode(), None))
    finally:
        dist.destroy_process_group()


def test_rccl_on_distinct_devices(oracle, hip_ctx, hip_lib):
    """The library's own RCCL transport with one rank per GPU (skipped on a one-GPU box)."""
    import ctypes
    from lambdaworks_cairo_prover_amd import api
    n = ctypes.c_int(0)
    hip_lib.sp_device_count(ctypes.byref(n))
    if n.value < 2:
        pytest.skip("needs at least two GPUs")
    world = 2 if n.value < 4 else 4
    fib_index, options = 2000, (4, 8, 3, 4)
    run = api.CairoRun.fibonacci(fib_index)
    want = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    assert want == oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, fib_index, options, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()               # exactly the processes started above
                p.join(timeout=30)
    for r, proof, stats in got:
        assert proof == want, (r, proof[:300])
        assert stats["world"] == world and stats["alltoall_calls"] >= 3


def _rccl_single_worker(port, q):
    import sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        ctx = api.Context(device=0)
        ctx.init_rccl()
        ctx.comm_selftest(1 << 20)
        ctx.comm_selftest(8)
        run = api.CairoRun.fibonacci(100)
        proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(4, 3, 3, 1))
        q.put(("ok", proof))
        ctx.close()
    except Exception:
        import traceback
        q.put(("fail", traceback.format_exc().encode()))
    finally:
        dist.destroy_process_group()


def test_rccl_transport_single_rank(oracle):
    """The library's RCCL communicator on the one GPU of the test box (world 1): communicator set-up, ncclAllGather and the
    grouped ncclSend / ncclRecv all-to-all run and deliver the documented layout (sp_comm_selftest)."""
    from lambdaworks_cairo_prover_amd import api
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_worker, args=(_free_port(), q))
    p.start()
    status, proof = q.get(timeout=600)
    p.join(timeout=60)
    assert status == "ok", proof[:2000]
    run = api.CairoRun.fibonacci(100)
    assert proof == oracle.cairo_prove(run.main_trace(), run.public_inputs_c, (4, 3, 3, 1))


def _air_worker(rank, world, port, n, options, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import wide_air
    from lambdaworks_cairo_prover_amd import api
    from test_wide_air import to_bytes
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = api.Context(device=0)
        ctx.set_collective(world, rank, api.StagedAllGather())
        ctx.set_option(api.SP_OPT_FRI_SHARD_MIN_LOG, 5)
        desc, keep = wide_air.build(n).build()
        proof = ctx.air_prove(desc, to_bytes(wide_air.main_trace(n)), api.ProofOptions(*options))
        q.put((rank, proof))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, ("fail: " + traceback.format_exc()).encode()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,options", [(2, 64, (4, 3, 3, 1)), (4, 256, (4, 4, 3, 2)), (2, 128, (8, 3, 3, 1))])
def test_sharded_program_air(world, n, options, oracle):
    """sp_air_prove on several ranks (five frame rows, caller-built auxiliary trace): the oracle's bytes."""
    import oracle_lib as O
    import wide_air
    from test_wide_air import to_bytes
    desc, keep = wide_air.build(n).build()
    want = O.program_air_prove(desc, to_bytes(wide_air.main_trace(n)), options)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_air_worker, args=(r, world, port, n, options, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert got[r] == want, (r, got[r][:400])


def test_link_rate_decides_the_interpolation_mode_null_transport():
    """Mode 2 of SP_OPT_SHARD_INTERPOLATION on rank 0's share of 8 ranks (timing-only transport): the assumed 46 GB/s interpolates
    everywhere, a stated faster fabric (SP_OPT_LINK_GBS) shards by column, and sp_comm_measure's figures - here of the null
    transport, i.e. of HBM memsets - replace the assumption when the caller stated nothing."""
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(2000)          # 2^14 rows
    opt = api.ProofOptions(8, 4, 3, 2)
    with api.Context(device=0) as ctx:
        ctx.init_null(8, 0)
        assert ctx.comm_measure(0)["allgather_gbs_per_link"] == 0.0          # nothing measured yet
        ctx.cairo_prove_run(run, opt)
        assert ctx.last_proof_info()["interpolation_sharded"] == 0
        m = ctx.comm_measure(64 << 20)                                       # "links" that are memsets: far beyond the threshold
        assert m["world"] == 8 and m["allgather_ms"] > 0 and m["alltoall_ms"] > 0
        expect = api.model_shard_interpolation(m["allgather_gbs_per_link"] / api.LINK_MEASURED_MARGIN, 8, 14)      # a measured rate decides with a margin
        ctx.cairo_prove_run(run, opt)
        assert ctx.last_proof_info()["interpolation_sharded"] == expect == 1, m
        ctx.set_option(api.SP_OPT_LINK_GBS, 46)                              # the caller's word beats the measurement
        ctx.cairo_prove_run(run, opt)
        assert ctx.last_proof_info()["interpolation_sharded"] == 0
        ctx.set_option(api.SP_OPT_LINK_GBS, 10000)
        ctx.cairo_prove_run(run, opt)
        assert ctx.last_proof_info()["interpolation_sharded"] == 1


@pytest.mark.parametrize("world,knobs", [(2, {"measure": True}), (4, {"measure": True, "async": True})], ids=["blocking-hooks", "stream-ordered-hooks"])
def test_link_rate_measurement_agrees_across_ranks(world, knobs, oracle, hip_ctx):
    """sp_comm_measure over transports that really move bytes between processes (the host-staged gloo hooks: blocking, and with the
    stream-ordered forms installed): every rank ends up with the SAME figures - the minimum over ranks, exchanged through the transport
    itself - so every rank takes the same interpolation-mode decision from them; the proof that follows has the single-rank bytes."""
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(300)
    options = (4, 4, 3, 2)
    want = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    results = _run_world(world, FIB(300), options, knobs)
    links = []
    for r in range(world):
        proof, stats = results[r]
        assert proof == want, (r, proof[:300])
        links.append(stats["link"])
    first = links[0]
    assert first["world"] == world and first["bytes_per_rank"] == 1 << 16
    assert first["allgather_gbs_per_link"] > 0 and first["alltoall_gbs_per_link"] > 0
    for r in range(world):           # sp_comm_time_ms: the blocking hooks' wall time, and - with the stream-ordered ones installed - event time
        stream_ms, blocking_ms = results[r][1]["comm_ms"]
        assert blocking_ms > 0 and (stream_ms > 0) == bool(knobs.get("async")), (r, stream_ms, blocking_ms)
    for other in links[1:]:          # the rates (a minimum over ranks) are identical everywhere; the local milliseconds need not be
        assert other["allgather_gbs_per_link"] == first["allgather_gbs_per_link"]
        assert other["alltoall_gbs_per_link"] == first["alltoall_gbs_per_link"]
        assert stats["interpolation_sharded"] == results[0][1]["interpolation_sharded"]


@pytest.mark.parametrize("world,knobs", [(2, {"measure": True, "measure_fault": 1}), (4, {"measure": True, "async": True, "measure_fault": 2})],
                         ids=["blocking-hooks", "stream-ordered-hooks"])
def test_link_rate_measurement_fails_on_every_rank_or_on_none(world, knobs, oracle, hip_ctx):
    """a rank that cannot prepare its measurement (out of memory for the payload, an event that cannot be created) used to
    return before its peers' first collective - they blocked, or paired that all-gather with the failing rank's next one, and the ranks
    ended up with different link rates, hence possibly different interpolation modes.  Now everything local is done first, a status word
    per rank goes round on every path, and ONE failing rank makes EVERY rank skip the timed collectives, drop the figures (a good earlier
    measurement included) and return non-zero.  That the transport stayed in step is shown by what follows on it: a proof with the
    single-rank bytes from every rank."""
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(300)
    options = (4, 4, 3, 2)
    want = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    results = _run_world(world, FIB(300), options, knobs)
    for r in range(world):
        proof, stats = results[r]
        assert proof == want, (r, proof[:300])
        link = stats["link"]
        assert link["raised"] is not None, (r, link)                          # every rank was told
        code, msg = link["raised"]
        assert code == (api._lib.SP_E_ALLOC if r == knobs["measure_fault"] else api._lib.SP_E_STATE), (r, link)
        if r != knobs["measure_fault"]:
            assert f"rank {knobs['measure_fault']} could not prepare" in msg, msg
        assert link["allgather_gbs_per_link"] == 0.0 and link["alltoall_gbs_per_link"] == 0.0 and link["world"] == 0, (r, link)
        assert stats["interpolation_sharded"] == 0                            # the assumed 46 GB/s on every rank


def test_comm_log_names_the_mode_and_the_rate(oracle, hip_ctx, capfd, monkeypatch):
    """SP_COMM_LOG ("always log the chosen mode and rate"): every rank prints, once per set-up shape, which interpolation mode
    it took, from which link rate (stated / measured / 1.25 / assumed) and where the threshold is - the line bench.py carries into its
    result for N > 1, so that the first run on real links can be read."""
    from lambdaworks_cairo_prover_amd import api
    monkeypatch.setenv("SP_COMM_LOG", "1")
    run = api.CairoRun.fibonacci(300)
    options = (4, 4, 3, 2)
    want = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    results = _run_world(2, FIB(300), options, {"measure": True})
    assert all(results[r][0] == want for r in range(2))
    err = capfd.readouterr().err
    lines = [l for l in err.splitlines() if l.startswith("[stark252 rank")]
    assert {l.split("]")[0] for l in lines} == {"[stark252 rank 0/2", "[stark252 rank 1/2"}, err[-800:]
    for l in lines:
        assert "interpolation on every rank" in l and "measured all-gather rate / 1.25" in l and "by column pays above" in l, l
