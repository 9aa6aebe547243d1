#!/usr/bin/env python3
"""Round 6: do a VALU-bound transform batch and a VALU-bound Keccak tree fill each other's bubbles when they run on two streams at once?
The passes keep the VALU 0.81 - 0.83 busy, leaf hashing 0.92: both are bound by the same unit, but their idle phases (tile loads / stores /
barriers; row loads) differ.  Two contexts (one stream each) on one device: a batch of size-2^20 transforms (what one coset of a trace
segment costs) and a tree over 2^23 leaves x 34 elements, alone and together.
usage: overlap_ntt_keccak.py [batch=34*8] [log_n=20] [log_leaves=22] [width=34]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from lambdaworks_cairo_prover_amd import api

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 34 * 8
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
log_leaves = int(sys.argv[3]) if len(sys.argv) > 3 else 22
width = int(sys.argv[4]) if len(sys.argv) > 4 else 34
dev = torch.device("cuda:0")
n, leaves = 1 << log_n, 1 << log_leaves
vec = torch.randint(0, 2**31 - 1, (batch, n, 8), dtype=torch.int32, device=dev)
vec[..., 7] &= 0x07FFFFFF
cols = torch.randint(0, 2**31 - 1, (width, leaves, 8), dtype=torch.int32, device=dev)
cols[..., 7] &= 0x07FFFFFF
nodes = torch.empty((2 * leaves - 1, 32), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
a, b = api.Context(device=0), api.Context(device=0)


def run_ntt(reps):
    for _ in range(reps):
        a.ntt_dev(vec.data_ptr(), n, batch)
    a.sync()


def run_tree(reps):
    for _ in range(reps):
        b.merkle_build_dev(cols.data_ptr(), leaves, width, leaves, nodes.data_ptr())
    b.sync()


def timed(fn, *args):
    t0 = time.perf_counter()
    fn(*args)
    return (time.perf_counter() - t0) * 1e3


run_ntt(3); run_tree(3)                          # tables, clocks
for rep in range(3):
    t_ntt = timed(run_ntt, 4) / 4
    t_tree = timed(run_tree, 4) / 4
    # reps chosen so that both streams are busy for about the same time
    r_ntt, r_tree = 4, max(1, round(4 * t_ntt / t_tree))
    alone = r_ntt * t_ntt + r_tree * t_tree
    th = threading.Thread(target=run_tree, args=(r_tree,))
    t0 = time.perf_counter()
    th.start(); run_ntt(r_ntt); th.join()
    both = (time.perf_counter() - t0) * 1e3
    print(f"transforms {batch} x 2^{log_n}: {t_ntt:8.3f} ms   tree 2^{log_leaves} x {width}: {t_tree:8.3f} ms   "
          f"{r_ntt} + {r_tree} one after the other {alone:8.2f} ms, on two streams {both:8.2f} ms  ({alone / both:.3f} x)")
a.close(); b.close()
