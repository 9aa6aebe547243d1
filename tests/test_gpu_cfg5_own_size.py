"""BASELINE configs[4] AT ITS OWN SIZE - trace 2^24 rows x 52 columns, blowup 16, 8 ranks (two LDE cosets each), 80 queries, 20-bit
grinding - through the sharded prover: the eight ranks run one at a time on the one GPU of the box, their 36 collectives (16 digest
all-to-alls of up to 1 GB per rank, 20 all-gathers; 43 GB in all) replayed from host recordings (tools/replay_ranks.py; the harness
itself is pinned to the CPU oracle's bytes at small sizes by tests/test_gpu_replay_ranks.py).  One proof comes out of all eight ranks;
the host verifier - the one that accepts the reference's own golden file - accepts it and rejects it after a byte flip; its sha256 is
the one recorded when the same shape was run with replicated AND with by-column interpolation and with Keccak trees
(profiles/r05_cfg5_replayed_proof*.txt).  The CPU oracle cannot follow here (0.5 TB); 169 GB of HBM, 45 GB of host memory, ~3 min."""
import ctypes
import hashlib
import os
import sys

import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

SHA_CFG5_KECCAK = "8245f2c61c3ca69f66ef5f716a91853a226cfe3e9fd1ac9fdefcf9b4cc3728e6"     # profiles/r05_cfg5_replayed_proof.txt


def _free_device_bytes():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipSetDevice(0) == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _available_host_bytes():
    limit = None
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        limit = None if v == "max" else int(v)
    except OSError:
        pass
    avail = None
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            avail = int(line.split()[1]) * 1024
    return min(x for x in (limit, avail) if x is not None)


def test_configs4_at_its_own_size_eight_ranks_replayed_on_one_gpu():
    from conftest import session_elapsed_s
    if session_elapsed_s() > 900 and os.environ.get("SP_RUN_OWN_SIZE") is None:
        pytest.skip("the suite has been running for more than 15 minutes on this box (a contended host): the four-minute own-size replay "
                    "is left to `python tools/replay_ranks.py` / SP_RUN_OWN_SIZE=1 rather than risk the wall-clock limit of the whole run")
    if _free_device_bytes() < 200e9:
        pytest.skip("needs 200 GB of free device memory")
    if _available_host_bytes() < 80e9:
        pytest.skip("needs 80 GB of host memory for the recorded collectives")
    from replay_ranks import sharded_proof_by_replay
    run = api.CairoRun.fibonacci(2389960)
    assert run.n_rows == 1 << 24
    opt = api.ProofOptions(16, 80, 3, 20)
    with api.Context(device=0) as ctx:
        proofs, stats = sharded_proof_by_replay(api, ctx, lambda c: c.cairo_prove_run(run, opt), 8, log=lambda *_: None)
        info, device_bytes = ctx.last_proof_info(), ctx.prover_device_bytes()
    assert sorted(proofs) == list(range(8)) and len({hashlib.sha256(p).hexdigest() for p in proofs.values()}) == 1
    assert hashlib.sha256(proofs[0]).hexdigest() == SHA_CFG5_KECCAK
    assert stats["collectives"] == 36 and stats["alltoalls"] == 16 and info["groups"] == 8 and info["fri_sharded_layers"] == 13
    assert 150e9 < device_bytes < 200e9
    assert api.cairo_verify(proofs[0], run.public_inputs_c, opt)
    bad = bytearray(proofs[0])
    bad[len(bad) // 3] ^= 1
    assert not api.cairo_verify(bytes(bad), run.public_inputs_c, opt)
    assert not api.cairo_verify(proofs[0], run.public_inputs_c, api.ProofOptions(8, 80, 3, 20))
