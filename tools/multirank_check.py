#!/usr/bin/env python3
"""Sharded prover at a larger size: `world` ranks on one GPU (gloo-staged all-gather) vs the single-rank proof.
usage: multirank_check.py <fib_index> <blowup> <world>"""
import hashlib, os, socket, sys, time
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, fib, blowup, q):
    import torch.distributed as dist
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run = api.CairoRun.fibonacci(fib)
    ctx = api.Context(device=0)
    ctx.set_collective(world, rank, api.StagedAllGather())
    t0 = time.time()
    proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(blowup, 10, 3, 8))
    q.put((rank, hashlib.sha256(proof).hexdigest(), time.time() - t0))
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    fib, blowup, world = (int(x) for x in sys.argv[1:4])
    from lambdaworks_cairo_prover_amd import api
    run = api.CairoRun.fibonacci(fib)
    ctx = api.Context(device=0)
    single = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(blowup, 10, 3, 8))
    ok = api.cairo_verify(single, run.public_inputs_c, api.ProofOptions(blowup, 10, 3, 8))
    ctx.close()
    print("single-rank:", hashlib.sha256(single).hexdigest(), "rows", run.n_rows, "verifies", ok, flush=True)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    c = mp.get_context("spawn"); q = c.Queue()
    ps = [c.Process(target=worker, args=(r, world, port, fib, blowup, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=900) for _ in ps]
    [p.join() for p in ps]
    for r in sorted(res):
        print(r)
    assert all(r[1] == hashlib.sha256(single).hexdigest() for r in res), "MISMATCH"
    print("OK: identical proof bytes on", world, "ranks")
