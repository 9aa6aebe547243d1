#!/usr/bin/env python3
"""The PCIe-inclusive rate of the fine-grained host-buffer entry point: sp_ntt on a 2^22-element vector in host memory (upload + decode +
transform + encode + download, synchronous) beside sp_ntt_dev on the same vector resident in HBM - the figure DESIGN.md section 8 quotes next to
the bench metric (which never includes PCIe).  usage: ntt_host_rate.py [log_n=22] [reps=10]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << k
x = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)
x[:, 0] &= 0x07
pinned = torch.from_numpy(x.copy()).pin_memory().numpy()
with api.Context() as ctx:
    for name, buf in (("pageable", x), ("page-locked", pinned)):
        import ctypes
        call = lambda: api.check(ctx._lib.sp_ntt(ctx._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_uint64(n), 0, None))   # the C entry point, in place (api.Context.ntt copies its argument first)
        call()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
        ms = statistics.median(ts)
        print(f"sp_ntt 2^{k} from a {name} host buffer: median {ms:.2f} ms, min {min(ts):.2f} ms -> {(n // 2) * k / (ms * 1e-3):.3e} butterflies/s, "
              f"{64 * n / (ms * 1e-3) / 1e9:.1f} GB/s of the 64 B/element that cross PCIe")
    d = torch.from_numpy(api.fe_to_device(x)).cuda()
    for _ in range(50):
        ctx.ntt_dev(d.data_ptr(), n)
    ctx.timer_start()
    for _ in range(50):
        ctx.ntt_dev(d.data_ptr(), n)
    ms = ctx.timer_stop() / 50
    print(f"sp_ntt_dev 2^{k} resident in HBM: {ms:.3f} ms -> {(n // 2) * k / (ms * 1e-3):.3e} butterflies/s")
