#!/usr/bin/env python3
"""Where the host side of an upload lives: NUMA nodes of the box, the GPU's node, the nodes of the pages of a row-major table built
by the front-end (pageable) and of a page-locked allocation of the library, the CPUs the process may use.  usage: numa_probe.py"""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
libc = ctypes.CDLL(None, use_errno=True)
def page_nodes(addr, nbytes, samples=64):
    """NUMA node of `samples` pages of [addr, addr + nbytes) through move_pages(2) with nodes = NULL (query only)."""
    step = max(4096, (nbytes // samples) & ~4095)
    pages = [(addr & ~4095) + i * step for i in range(samples) if i * step < nbytes]
    arr = (ctypes.c_void_p * len(pages))(*pages)
    status = (ctypes.c_int * len(pages))()
    rc = libc.syscall(279, 0, ctypes.c_ulong(len(pages)), arr, None, status, 0)
    if rc != 0:
        return f"move_pages failed (errno {ctypes.get_errno()})"
    hist = {}
    for s in status:
        hist[s] = hist.get(s, 0) + 1
    return hist
for node in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    cpus = open(node + "/cpulist").read().strip()
    mem = [l for l in open(node + "/meminfo") if "MemTotal" in l or "MemFree" in l]
    print(os.path.basename(node), "cpus", cpus, " ".join(x.split(":")[1].strip() for x in mem))
for dev in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    print(dev, open(dev).read().strip())
print("affinity:", len(os.sched_getaffinity(0)), "CPUs; cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
run = api.CairoRun.fibonacci(149000)
tr = run.main_trace()
print("row-major table (numpy, built by sp_cairo_run_main_trace's threads):", page_nodes(tr.ctypes.data, tr.nbytes))
ptr, n, cols, pinned = run.columns()
print("run columns (page-locked:", pinned, "):", page_nodes(ptr, n * cols * 32))
fresh = np.ones(1 << 28, dtype=np.uint8)
print("fresh numpy array touched by this thread:", page_nodes(fresh.ctypes.data, fresh.nbytes), "running on cpu", libc.sched_getcpu())
opt = api.ProofOptions(8, 80, 3, 20)
import time
for _ in range(3):
    ctx.cairo_prove(tr, run.public_inputs_c, opt)
t0 = time.perf_counter(); ctx.cairo_prove(tr, run.public_inputs_c, opt); print(f"rows proof {1e3 * (time.perf_counter() - t0):.1f} ms", ctx.last_upload_stats())
