"""`python bench.py --gpus N` must not be able to hang or to end without its ONE JSON line: the launcher watches
every rank, a rank that dies or never joins ends the job within the stated timeouts, the line carries the reason, the code is non-zero.
No GPU needed: the injected fault strikes before the rendezvous, and nothing touches a device until the rendezvous has succeeded."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "config", "error"}


@pytest.fixture(scope="module", autouse=True)
def torch_paged_in():
    """The very first `import torch` of a fresh container takes one to two minutes (the image pages in); the ranks below import it too.
    Paying that once here keeps the time bounds of the cases about the launcher, not about the file cache."""
    import torch  # noqa: F401


def run_bench(extra_env, *argv, timeout=600):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.monotonic() - t0


def the_one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-1500:]
    return json.loads(lines[0])


def test_a_rank_that_exits_before_the_rendezvous_ends_the_job_quickly():
    r, took = run_bench({"SP_BENCH_FAULT_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline")
    line = the_one_line(r.stdout)
    assert r.returncode != 0 and took < 240, (r.returncode, took)
    assert KEYS <= set(line) and line["value"] is None and line["n_gpus"] == 2
    assert "rank 1 exited with code 3" in line["error"] or "terminated" in line["error"], line["error"]


def test_four_ranks_one_missing():
    r, took = run_bench({"SP_BENCH_FAULT_RANK": "3"}, "--gpus", "4", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline")
    line = the_one_line(r.stdout)
    assert r.returncode != 0 and took < 240 and line["value"] is None and line["n_gpus"] == 4


def test_a_rank_that_hangs_before_the_rendezvous_is_timed_out():
    """Rank 1 sleeps for ever: the others' rendezvous gives up after SP_BENCH_INIT_TIMEOUT_S, rank 0 prints the reason, the launcher
    terminates the sleeper."""
    r, took = run_bench({"SP_BENCH_FAULT_RANK": "1", "SP_BENCH_FAULT": "hang", "SP_BENCH_INIT_TIMEOUT_S": "15"},
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline")
    line = the_one_line(r.stdout)
    assert r.returncode != 0 and took < 400, (r.returncode, took)
    assert line["value"] is None and "rendezvous" in line["error"], line


def test_the_overall_deadline_ends_a_job_that_never_finishes():
    """Same sleeper, but the rendezvous timeout is long: the ranks' own deadline (SP_BENCH_DEADLINE_S) fires on a timer thread."""
    r, took = run_bench({"SP_BENCH_FAULT_RANK": "1", "SP_BENCH_FAULT": "hang", "SP_BENCH_INIT_TIMEOUT_S": "600", "SP_BENCH_DEADLINE_S": "12"},
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline")
    line = the_one_line(r.stdout)
    assert r.returncode != 0 and took < 400, (r.returncode, took)
    assert line["value"] is None and "deadline" in line["error"], line


def test_under_torchrun_rank_0_still_prints_its_line_when_a_peer_dies():
    """The driver's own command line: torch.distributed.run terminates the other workers when one fails; rank 0's SIGTERM handler (or its
    rendezvous timeout) leaves the JSON line on stdout."""
    env = dict(os.environ, SP_BENCH_FAULT_RANK="1", SP_BENCH_INIT_TIMEOUT_S="20")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    took = time.monotonic() - t0
    assert r.returncode != 0 and took < 400
    line = the_one_line(r.stdout)
    assert line["value"] is None and line["n_gpus"] == 2 and line["error"]


def test_a_failure_behind_the_rendezvous_still_leaves_the_line():
    """No fault injected: on a machine without a GPU both ranks meet, then fail when they touch the device - rank 0's line names the
    stage and the exception, the code is non-zero (on a GPU box the same command simply succeeds: tests/test_gpu_bench_multirank.py)."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the command succeeds here")
    r, took = run_bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline")
    line = the_one_line(r.stdout)
    assert r.returncode != 0 and took < 400
    assert line["value"] is None and line["n_gpus"] == 2 and "failed in stage 'setup'" in line["error"], line


def test_sigterm_to_the_launcher_ends_the_ranks_and_leaves_the_line():
    """The launcher is told to stop (a driver's timeout sends SIGTERM): it terminates its ranks - they sit in process groups of their own -
    prints the one line and returns non-zero; and no rank outlives it (PR_SET_PDEATHSIG covers even a SIGKILL of the launcher)."""
    import signal
    env = dict(os.environ, SP_BENCH_FAULT_RANK="1", SP_BENCH_FAULT="hang", SP_BENCH_INIT_TIMEOUT_S="600")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--proof", "0", "--no-cpu-baseline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8.0)                      # both ranks are up: rank 1 sleeps, rank 0 waits for it in the rendezvous
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 2, kids
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=240)
    line = the_one_line(out)
    assert p.returncode != 0 and line["value"] is None and "signal" in (line["error"] + str(line.get("launcher_note", ""))), line
    time.sleep(0.5)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", k


def test_children_of_a_rank_arm_the_parent_death_signal_themselves():
    """the proof and pre-flight children are started by a rank that has initialised torch, HIP and gloo - a preexec_fn (Python
    between fork and exec of a multi-threaded process) can deadlock there.  _spawn_rank_child passes no preexec_fn; the child arms
    PR_SET_PDEATHSIG as its first statement and leaves at once when the process named in SP_BENCH_PARENT_PID is not its parent any more;
    a child whose parent is then KILLED goes with it."""
    import inspect
    import signal
    sys.path.insert(0, ROOT)
    import bench
    src = inspect.getsource(bench._spawn_rank_child)
    assert "preexec_fn=" not in src and "SP_BENCH_PARENT_PID" in src
    # (1) the named parent is gone already: exit code 86 before anything is imported or any device is touched
    code = "import sys; sys.path.insert(0, %r); import bench; bench._arm_parent_death_signal(); print('armed')" % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SP_BENCH_PARENT_PID="1", SP_BENCH_PDEATHSIG=str(int(signal.SIGKILL))),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 86 and "armed" not in r.stdout, (r.returncode, r.stdout, r.stderr[-300:])
    # (2) the right parent: armed; then the parent is killed outright and the child is gone within moments
    parent = ("import os, subprocess, sys, time\n"
              "child = subprocess.Popen([sys.executable, '-c', %r], env=dict(os.environ, SP_BENCH_PARENT_PID=str(os.getpid()), SP_BENCH_PDEATHSIG='9'))\n"
              "print(child.pid, flush=True)\n"
              "time.sleep(600)\n") % ("import sys, time; sys.path.insert(0, %r); import bench; bench._arm_parent_death_signal(); print('armed', flush=True); time.sleep(600)" % ROOT)
    p = subprocess.Popen([sys.executable, "-c", parent], stdout=subprocess.PIPE, text=True)
    try:
        kid = int(p.stdout.readline())
        assert p.stdout.readline().strip() == "armed"
        p.kill()
        p.wait()
        for _ in range(100):
            if not os.path.exists(f"/proc/{kid}") or open(f"/proc/{kid}/stat").read().split()[2] == "Z":
                break
            time.sleep(0.1)
        else:
            os.kill(kid, signal.SIGKILL)
            raise AssertionError("the child outlived its killed parent")
    finally:
        if p.poll() is None:
            p.kill()
