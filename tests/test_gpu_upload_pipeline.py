"""The row-major upload pipeline (csrc/prover_upload.cpp: gather threads, ring, column groups, flag bitmaps) on SMALL tables.

The library takes that path from 64 MB of table on (SP_UPLOAD_MIN_MB, read once per process), so the ordinary parity tests reach it
only with the big 34-column traces.  Here child processes set the threshold to 0 and prove small traces through sp_cairo_prove
(reference prover.rs:532-541: `prove(&TraceTable)`), for every group layout the switches offer, against the CPU oracle's bytes:
34 and 43 columns (range-check builtin: 27 full-width columns behind the 16 flags, a three-column tail group), 64 rows (one bitmap
word per column) and 32 rows (no bitmaps: n is not a multiple of 64), a flag cell that is not a bit (the upload is repeated in full),
a trace that violates a constraint, and both field encodings."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
from test_rc_builtin import run_of
out = []
def case(name, ctx, trace, pub, options, expect_kind="row-major host buffer, gathered by host threads"):
    import time
    t0 = time.perf_counter()
    want = oracle.cairo_prove(trace, pub, options)
    t1 = time.perf_counter()
    got = ctx.cairo_prove(trace, pub, api.ProofOptions(*options))
    t2 = time.perf_counter()
    st = ctx.last_upload_stats()
    out.append({"case": name, "oracle_s": round(t1 - t0, 2), "device_s": round(t2 - t1, 2), "same_bytes": got == want, "kind": st["kind"], "kind_ok": st["kind"] == expect_kind, "groups": st["groups"], "bytes": st["bytes"]})
with api.Context(device=0) as ctx:
    for fib, options in ((100, (4, 3, 3, 1)), (10, (2, 3, 3, 1)), (3, (4, 3, 3, 1)), (300, (8, 5, 3, 2)), (4000, (2, 3, 3, 1))):      # fib(3): 64 rows; fib(4000): 2^15 rows, several chunks a group at SP_UPLOAD_CHUNK_MB=1
        run = api.CairoRun.fibonacci(fib)
        case(f"fib({fib})", ctx, run.main_trace(), run.public_inputs_c, options)
    run = api.CairoRun.fibonacci(1)                      # 32 rows: not a multiple of 64, no bitmaps
    tr = run.main_trace()
    if tr.shape[0] < 64:
        case("32 rows", ctx, tr, run.public_inputs_c, (4, 3, 3, 1))
    for name, options in (("rc_program", (4, 3, 3, 1)), ("rc_loop_300", (2, 4, 3, 1)), ("output_and_rc", (8, 3, 3, 1))):
        run = run_of(name)
        assert run.main_trace().shape[1] == 43
        case(name, ctx, run.main_trace(), run.public_inputs_c, options)
    run = api.CairoRun.fibonacci(100)
    tr = run.main_trace().copy()
    tr[tr.shape[0] - 1, 15, 31] = 2                      # the last cell of the last flag column is not a bit: uploaded again in full
    case("flag cell 2", ctx, tr, run.public_inputs_c, (4, 3, 3, 1))
    tr = run.main_trace().copy()
    tr[5, 20, 31] ^= 1                                    # a full-width cell: constraints violated, the whole-domain composition
    case("violated", ctx, tr, run.public_inputs_c, (4, 3, 3, 1))
with api.Context(device=0, fe_encoding=api.SP_FE_MONT_LIMBS) as ctx:      # lambdaworks' in-memory limbs through the same pipeline
    run = api.CairoRun.fibonacci(100)
    tr_be = run.main_trace()
    want = oracle.cairo_prove(tr_be, run.public_inputs_c, (4, 3, 3, 1))
    got = ctx.cairo_prove(run.main_trace(fe_encoding=api.SP_FE_MONT_LIMBS), run.public_inputs_c, api.ProofOptions(4, 3, 3, 1))
    out.append({"case": "lambdaworks limbs", "same_bytes": got == want, "kind": ctx.last_upload_stats()["kind"], "kind_ok": True, "groups": ctx.last_upload_stats()["groups"], "bytes": 0})
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("env", [{}, {"SP_UPLOAD_ORDER": "seq"}, {"SP_UPLOAD_PACKW": "1"}, {"SP_UPLOAD_PACKW": "3"}, {"SP_UPLOAD_PACKW": "2", "SP_UPLOAD_MAXW": "4"},
                                 {"SP_UPLOAD_NO_FLAG_PACK": "1"}, {"SP_UPLOAD_CHUNK_MB": "1"}],
                         ids=lambda e: ",".join(f"{k[10:]}={v}" for k, v in e.items()) or "default")
def test_small_tables_through_the_upload_pipeline(env):
    e = dict(os.environ, SP_UPLOAD_MIN_MB="0", **env)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    res = json.loads(lines[-1][7:])
    assert len(res) >= 11
    bad = [c for c in res if not (c["same_bytes"] and c["kind_ok"])]
    assert not bad, bad


RESHAPE_CHILD = r'''
import json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
out = []
with api.Context(device=0) as ctx:
    # one context, shapes that change with equal or fewer rows: the flag-bitmap staging buffer must be re-carved with the shape
    # (ADVICE r4: it used to survive free_all() and point into freed / re-used arena memory)
    for fib, options in ((4000, (2, 3, 3, 1)), (4000, (8, 3, 3, 1)), (2000, (4, 3, 3, 1)), (4000, (4, 3, 3, 1)), (300, (2, 3, 3, 1)), (4000, (2, 3, 3, 1))):
        run = api.CairoRun.fibonacci(fib)
        want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
        got = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options))
        st = ctx.last_upload_stats()
        out.append({"case": f"fib({fib}) blowup {options[0]}", "same_bytes": got == want, "kind": st["kind"]})
print("RESULT " + json.dumps(out))
'''


def test_shape_changes_on_one_context_with_flag_bitmaps():
    e = dict(os.environ, SP_UPLOAD_MIN_MB="0")
    r = subprocess.run([sys.executable, "-c", RESHAPE_CHILD, ROOT], env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    res = json.loads(lines[-1][7:])
    assert len(res) == 6
    bad = [c for c in res if not (c["same_bytes"] and c["kind"].startswith("row-major host buffer, gathered"))]
    assert not bad, bad


BUDGET_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib as oracle
from lambdaworks_cairo_prover_amd import api
out = {"cpus": api.host_cpus(), "budget": api.host_cpu_budget()}
run = api.CairoRun.fibonacci(4000)
options = (4, 5, 3, 2)
want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
with api.Context(device=0) as ctx:
    before = len(os.listdir("/proc/self/task"))
    out["rows"] = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options)) == want
    out["pool_threads"] = len(os.listdir("/proc/self/task")) - before
    out["run"] = ctx.cairo_prove_run(run, api.ProofOptions(*options)) == want
    ctx.set_option(api.SP_OPT_HOST_RANKS, 1)            # the option beats the environment; the pool follows at the next upload
    out["budget_after_option"] = api.host_cpu_budget()
    out["rows_again"] = ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(*options)) == want
    out["pool_threads_after_option"] = len(os.listdir("/proc/self/task")) - before
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("ranks", ["8", "1000"])
def test_per_host_thread_budget_on_the_rows_path(ranks):
    """SP_HOST_RANKS = 8: this rank's share of the CPUs sizes the gather pool; = 1000 (more ranks than CPUs): one worker, every wait
    blocks instead of polling (common.h sp_stream_wait_polling) and the orchestrating thread yields when idle - same bytes either way."""
    e = dict(os.environ, SP_UPLOAD_MIN_MB="0", SP_HOST_RANKS=ranks)
    e.pop("LOCAL_WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", BUDGET_CHILD, ROOT], env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    res = json.loads(lines[-1][7:])
    assert res["rows"] and res["run"] and res["rows_again"], res
    cpus = res["cpus"]
    assert res["budget"] == [max(1, cpus // int(ranks)), int(ranks)]
    want_pool = max(2, min(24, 2 * max(1, cpus // int(ranks)))) - 1
    assert res["pool_threads"] >= want_pool and res["pool_threads"] <= want_pool + 6, res        # (+ the runtime's own helper threads of a first proof)
    assert res["budget_after_option"] == [cpus, 1]
    assert res["pool_threads_after_option"] >= max(2, min(24, 2 * cpus)) - 1, res
