#!/usr/bin/env python3
"""Kernel timeline of the LAST proof in a rocprofv3 kernel trace (rocpd SQLite): start offset, duration, gap to the previous
kernel, grid size, name.  A proof starts at its first aux_prepare_kernel... no: at the first kernel after the previous proof's
gather_jobs_kernel.  usage: rocprof_timeline.py <kernel_trace.db> [which_proof_from_the_end=1]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if "gather_jobs_kernel" in r[0]]
    if len(ends) < back + 1:
        print("not enough proofs in the trace")
        return
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    seg = rows[lo:hi]
    t0 = seg[0][1]
    prev_end = t0
    busy = 0
    print(f"# proof = kernels {lo}..{hi - 1} of {sys.argv[1]}")
    print(f"{'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s} {'grid':>10s}  kernel")
    for name, s, e, grid in seg:
        short = name.split("(")[0][-44:]
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {(s - prev_end) / 1e3:8.1f} {grid:10d}  {short}")
        busy += e - s
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"# span {span:.1f} us, kernels busy {busy / 1e3:.1f} us ({100 * busy / 1e3 / span:.1f} %)")


if __name__ == "__main__":
    main()
