#!/usr/bin/env python3
"""Where a proof from the reference's row-major host table (sp_cairo_prove) loses time against the resident one: per call the wall
time, the device time of each round and the upload statistics, resident and row-major calls interleaved.
usage: rows_path_rounds.py [fib=149000] [blowup=8] [iterations=6] [near|far]
near / far: build the row-major table on the GPU's NUMA node / on the other one (two-socket hosts; default: wherever the threads run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
fib = int(sys.argv[1]) if len(sys.argv) > 1 else 149000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
where = sys.argv[4] if len(sys.argv) > 4 else ""
ctx = api.Context()
run = api.CairoRun.fibonacci(fib)
import ctypes, glob
libc = ctypes.CDLL(None, use_errno=True)
def node_of(addr, nbytes):
    pages = [(addr & ~4095) + i * max(4096, (nbytes // 16) & ~4095) for i in range(16)]
    arr = (ctypes.c_void_p * 16)(*pages); st = (ctypes.c_int * 16)()
    if libc.syscall(279, 0, ctypes.c_ulong(16), arr, None, st, 0) != 0: return -1
    return max(set(st), key=list(st).count)
def node_cpus(node):
    out = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
    return out
ptr, nr, nc, pinned = run.columns()
gpu_node = node_of(ptr, nr * nc * 32) if pinned else -1       # page-locked memory of the library lives on the GPU's node
nodes = sorted(int(os.path.basename(p)[4:]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
if where in ("near", "far") and gpu_node >= 0 and len(nodes) > 1:
    target = gpu_node if where == "near" else next(n for n in nodes if n != gpu_node)
    allowed = os.sched_getaffinity(0)
    os.sched_setaffinity(0, node_cpus(target) & allowed)       # the threads that fill the table inherit this
    tr = run.main_trace()
    os.sched_setaffinity(0, allowed)
else:
    tr = run.main_trace()
print(f"GPU on NUMA node {gpu_node}, row-major table on node {node_of(tr.ctypes.data, tr.nbytes)}")
opt = api.ProofOptions(b, 80, 3, 20)
dev = torch.from_numpy(tr).cuda(); torch.cuda.synchronize()
for _ in range(3):
    ctx.cairo_prove(tr, run.public_inputs_c, opt)
    ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt)
la = os.getloadavg()
print(f"host: {api.host_cpus()} usable CPUs, load average {la[0]:.1f} {la[1]:.1f} (of {os.cpu_count()} hardware threads)")
for it in range(iters):
    t0 = time.perf_counter(); ctx.cairo_prove_dev(dev.data_ptr(), tr.shape[0], tr.shape[1], run.public_inputs_c, opt); td = 1e3 * (time.perf_counter() - t0)
    rd = ctx.last_round_ms()
    t0 = time.perf_counter(); ctx.cairo_prove(tr, run.public_inputs_c, opt); th = 1e3 * (time.perf_counter() - t0)
    rh = ctx.last_round_ms(); s = ctx.last_upload_stats()
    print(f"[{it}] resident {td:6.1f} ms rounds {['%.1f' % x for x in rd[1:]]}   rows {th:6.1f} ms rounds {['%.1f' % x for x in rh[1:]]}   "
          f"gather {s['gather_gbs']} GB/s dma {s['dma_gbs']} exposed {s['exposed_ms']} max stall {s['max_stall_ms']} host loop {s['host_ms']} ms", flush=True)
