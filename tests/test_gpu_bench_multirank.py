"""`bench.py --gpus 2` on the GPU box, ranks sharing device 0: the whole N > 1 path of the driver's command - launcher, gloo control
plane with bounded barriers, the NTT measurement on every rank, the sharded proofs in child processes - with small programs, and the
RCCL pre-flight's failure branch for real: two ranks on ONE device cannot build an RCCL communicator (ncclCommInitRank refuses a
duplicate GPU), so the pre-flight child fails on every rank, the ranks agree to fall back to the host-staged hooks, and the proofs
still come out (the first multi-GPU run may not be lost, whatever the fabric does)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_with_a_failing_rccl_preflight_fall_back_to_staged_hooks():
    env = dict(os.environ, SP_BENCH_FORCE_DEVICE="0", SP_BENCH_TRANSPORT="rccl")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "16",
                        "--proof-fib", "2000", "--proof-blowup", "4", "--cfg4-fib", "1000", "--cfg4-blowup", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "error" not in line
    summ = line["summary"]
    assert summ["rccl"]["backend"] == "gloo-staged hook" and summ["rccl"]["world"] == 2 and summ["rccl"]["selftest"] == "ok"
    assert set(summ["rccl_preflight_failed"]) == {"0", "1"}                       # both ranks' pre-flight children failed, both said so
    for name in ("cfg3", "cfg4"):
        assert summ[name]["n_gpus"] == 2 and len(summ[name]["sha"]) == 8 and summ[name]["resident_ms"][0] > 0, summ[name]
