"""Coset-sharded prover: world_size 2 and 4 ranks (sharing the one GPU of the test box, exchanging through the gloo-staged
all-gather hook) produce the SAME proof bytes as the single-rank prover and the CPU oracle (SURVEY.md §8(e) gate)."""
import os
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


def _worker(rank, world, port, fib_index, options, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = api.CairoRun.fibonacci(fib_index)
        ctx = api.Context(device=0)
        ctx.set_collective(world, rank, api.StagedAllGather())
        proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
        proof2 = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))  # buffer reuse path
        q.put((rank, proof if proof == proof2 else b"MISMATCH-ON-REUSE"))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, ("fail: " + traceback.format_exc()).encode()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fib_index,options", [(2, 100, (4, 3, 3, 1)), (4, 100, (4, 5, 3, 2)), (4, 200, (8, 4, 3, 1)), (2, 60, (2, 3, 3, 1)),
                                                     (8, 100, (8, 4, 3, 1)), (8, 60, (16, 3, 3, 1))])
def test_sharded_proof_bytes_identical(world, fib_index, options, oracle, hip_ctx):
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(fib_index)
    want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    single = hip_ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
    assert single == want
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fib_index, options, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert results[r] == want, (r, results[r][:300])
