"""world_size-2 gloo test (CPU) of the multi-GPU protocol: coset sharding map, the all-gather hook and the reassembly of
leaf digests / evaluations into natural LDE order. The shard-local compute is stood in by the oracle (checker); on the GPU
box tests/test_gpu_multirank.py runs the real sharded prover through the same hook."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import ctypes
    import random
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import oracle_lib as O
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = random.Random(7)
        k, logb, h = 6, 2, 3
        n, b = 1 << k, 1 << logb
        shard_log = world.bit_length() - 1
        coeffs = api.felts_to_bytes([rng.randrange(api.P) for _ in range(n)])
        full = O.lde(coeffs, b, h)                                   # natural order, N = n*b
        # this rank's cosets, in local order q*b_loc + c_loc
        b_loc = b >> shard_log
        local_idx = [api.shard_global_index(i, logb, shard_log, rank) for i in range(n * b_loc)]
        local = np.ascontiguousarray(full[local_idx])
        # all-gather through the production hook (host-memory mode) and reassemble
        hook = api.StagedAllGather(device_memory=False)
        recv = np.empty((world,) + local.shape, dtype=np.uint8)
        rc = hook.cfn(None, local.ctypes.data, recv.ctypes.data, local.nbytes)
        assert rc == 0
        again = api.interleave_shards(recv, n, logb, shard_log)
        assert np.array_equal(again, full)
        # leaf digests exchanged instead of rows: tree over reassembled digests == tree over the full rows
        local_leaves = np.stack([np.frombuffer(O.keccak256(row.tobytes()), dtype=np.uint8) for row in local])
        recv_d = np.empty((world,) + local_leaves.shape, dtype=np.uint8)
        assert hook.cfn(None, local_leaves.ctypes.data, recv_d.ctypes.data, local_leaves.nbytes) == 0
        leaves = api.interleave_shards(recv_d, n, logb, shard_log)
        level = [bytes(x) for x in leaves]
        while len(level) > 1:
            level = [O.keccak256(level[i] + level[i + 1]) for i in range(0, len(level), 2)]
        assert level[0] == O.merkle_build(full.reshape(-1, 1, 32))
        # Merkle combine through the all-to-all hook: block d of the local digest array (local natural
        # order, leaf l = global leaf l*G + rank) is this rank's share of the contiguous leaf range of rank d
        N = n * b
        per = N // world // world
        send = np.ascontiguousarray(local_leaves)
        recv = np.empty_like(send)
        assert hook.a2a_cfn(None, send.ctypes.data, recv.ctypes.data, per * 32) == 0
        mine = api.interleave_shards(recv.reshape(world, per, 32), per, shard_log, shard_log)   # out[j*G + s] = recv[s][j]
        assert np.array_equal(mine, leaves[rank * (N // world):(rank + 1) * (N // world)])
        sub = [bytes(x) for x in mine]
        while len(sub) > 1:
            sub = [O.keccak256(sub[i] + sub[i + 1]) for i in range(0, len(sub), 2)]
        roots = np.empty((world, 32), dtype=np.uint8)
        mine_root = np.frombuffer(sub[0], dtype=np.uint8).copy()
        assert hook.cfn(None, mine_root.ctypes.data, roots.ctypes.data, 32) == 0
        top = [bytes(x) for x in roots]
        while len(top) > 1:
            top = [O.keccak256(top[i] + top[i + 1]) for i in range(0, len(top), 2)]
        assert top[0] == level[0]
        # frame rows i and i+b stay inside one coset: local index + b_loc
        for i in range(n * b_loc):
            nxt = (i + b_loc) % (n * b_loc)
            assert api.shard_global_index(nxt, logb, shard_log, rank) == (api.shard_global_index(i, logb, shard_log, rank) + b) % (n * b)
        # owner rule used by the query phase
        for g in range(n * b):
            owner = (g % b) % world
            if owner == rank:
                assert g in local_idx
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_coset_sharding_protocol_over_gloo(world, oracle, hip_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
