#!/usr/bin/env python3
"""Where the slow proofs of the row-major host-table path come from: consecutive sp_cairo_prove calls, each with its upload statistics,
the cgroup's throttling counters and the process' involuntary context switches; prints the slowest beside the median.
usage: rows_tail.py [fib=70000] [blowup=4] [iterations=150] [upload_threads=0 (default)]"""
import os, resource, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api


def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            continue
    return out


fib = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 150
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
api.host_bind_to_device(0)
ctx = api.Context()
if threads:
    ctx.set_option(api.SP_OPT_UPLOAD_THREADS, threads)
run = api.CairoRun.fibonacci(fib)
tr = run.main_trace()
opt = api.ProofOptions(b, 80, 3, 20)
for _ in range(5):
    ctx.cairo_prove(tr, run.public_inputs_c, opt)
rows = []
for it in range(iters):
    c0, r0 = cpu_stat(), resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    ctx.cairo_prove(tr, run.public_inputs_c, opt)
    ms = 1e3 * (time.perf_counter() - t0)
    c1, r1 = cpu_stat(), resource.getrusage(resource.RUSAGE_SELF)
    st = ctx.last_upload_stats()
    rows.append({"it": it, "ms": round(ms, 2), "round1": round(ctx.last_round_ms()[1], 2), "gather": st["gather_ms"], "host": st["host_ms"], "dma": st["dma_ms"],
                 "exposed": st["exposed_ms"], "stall": st["max_stall_ms"], "throttled": c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0),
                 "throttled_us": c1.get("throttled_usec", c1.get("throttled_time", 0)) - c0.get("throttled_usec", c0.get("throttled_time", 0)),
                 "nivcsw": r1.ru_nivcsw - r0.ru_nivcsw, "cpu_ms": round(1e3 * ((r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)), 1)})
v = sorted(r["ms"] for r in rows)
print(f"# fib {fib} blowup {b} threads {threads or 'default'}: {len(v)} proofs  min {v[0]:.1f}  median {statistics.median(v):.1f}  p95 {v[int(0.95 * len(v))]:.1f}  max {v[-1]:.1f} ms; "
      f"cpus {len(os.sched_getaffinity(0))}, cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '?'}")
by = sorted(rows, key=lambda r: r["ms"])
print("median:", by[len(by) // 2])
for r in by[-10:]:
    print("slow:  ", r)
