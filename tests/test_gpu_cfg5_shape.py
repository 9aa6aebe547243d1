"""BASELINE configs[4] in shape (blowup 16, eight ranks, two LDE cosets per rank), at the largest trace that lets the eight
ranks share the one GPU of the test box: n = 2^21 rows x 52 columns (configs[4] itself is n = 2^24 on eight GPUs).  The
sharded proof must equal the single-rank proof byte for byte and pass the library's verifier; the device memory each rank
holds is measured and extrapolated to n = 2^24 (it is linear in n) against the 288 GB of an MI355X.

About 40 s (every collective is staged through host memory and gloo here); needs ~200 GB of free device memory."""
import hashlib
import os
import socket

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

FIB_INDEX = 298000          # 2 086 009 steps -> 2^21 rows
OPTIONS = (16, 30, 3, 12)
WORLD = 8
HBM_BYTES = 288e9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from lambdaworks_cairo_prover_amd import api
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = api.CairoRun.fibonacci(FIB_INDEX)
        ctx = api.Context(device=0)
        ctx.set_collective(world, rank, api.StagedAllGather())
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, backend)
        proof = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*OPTIONS))
        q.put((rank, hashlib.sha256(proof).hexdigest(), ctx.prover_device_bytes(), ctx.last_proof_info(), ctx.comm_stats()))
        ctx.close()
    except Exception:
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0, None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", [0, 1], ids=["keccak256", "poseidon"])
def test_cfg5_shape_eight_ranks_two_cosets_each(hip_lib, backend):
    """backend 1: configs[4] names Poseidon Merkle trees - the optional backend (SP_OPT_MERKLE_BACKEND, no reference counterpart)."""
    import ctypes
    from lambdaworks_cairo_prover_amd import api
    hip = ctypes.CDLL("libamdhip64.so")          # the runtime the library itself uses (no second HIP runtime in this process)
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipSetDevice(0) == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    free = free.value
    if free < 200e9:
        pytest.skip("needs ~200 GB of free device memory")
    run = api.CairoRun.fibonacci(FIB_INDEX)
    assert run.n_rows == 1 << 21
    with api.Context(device=0) as ctx:                      # single rank: the reference bytes, then free its 80 GB
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, backend)
        single = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*OPTIONS))
        single_bytes = ctx.prover_device_bytes()
        single_ms = sum(ctx.last_round_ms())
    assert api.cairo_verify(single, run.public_inputs_c, api.ProofOptions(*OPTIONS), backend)
    assert not api.cairo_verify(single, run.public_inputs_c, api.ProofOptions(*OPTIONS), 1 - backend)
    want = hashlib.sha256(single).hexdigest()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, WORLD, port, backend, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=3000) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    per_rank = []
    for rank, digest, dev_bytes, info, stats in sorted(got):
        assert digest == want, (rank, digest[:400])
        # (interpolation: the link model's choice - on every rank at 46 GB/s per link; by column is pinned in test_gpu_multirank*.py)
        assert info["groups"] == WORLD and info["interpolation_sharded"] == 0 and info["composition_path"] == 1
        assert info["fri_sharded_layers"] >= 5          # layers of >= 2^16 leaves keep their evaluations and trees sharded
        assert stats["alltoall_calls"] >= 3 + info["fri_sharded_layers"]
        per_rank.append(dev_bytes)
    peak = max(per_rank)
    at_cfg5 = peak * 8                                  # n = 2^24 instead of 2^21: every buffer is linear in n
    print(f"\n[{'poseidon' if backend else 'keccak256'} trees, single-rank proof {single_ms:.0f} ms of device time] "
          f"per-rank device bytes at n = 2^21, blowup 16, world 8: {peak / 1e9:.1f} GB (single rank: {single_bytes / 1e9:.1f} GB); "
          f"extrapolated to configs[4] (n = 2^24): {at_cfg5 / 1e9:.0f} GB of {HBM_BYTES / 1e9:.0f} GB")
    assert at_cfg5 < 0.9 * HBM_BYTES
    assert peak < single_bytes / 3                      # sharding really divides the footprint
