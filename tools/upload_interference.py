#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/rows_only.py: durations of the per-group transform launches of round 1 while the upload
is still running (before the last rows_to_columns kernel) against the same launches after it.  usage: upload_interference.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "gather_jobs_kernel" in r[0]]
for back in (3, 2, 1):
    seg = rows[ends[-back - 1] + 1:ends[-back] + 1]
    first_leaf = next(i for i, r in enumerate(seg) if "leaf_hash" in r[0] and (r[2] - r[1]) > 2e6)
    r1 = seg[:first_leaf]
    last_r2c = max(r[2] for r in r1 if "rows_to_columns" in r[0])
    out = {}
    for n, s, e, g in r1:
        if "ntt_pass" not in n: continue
        key = (n.split("ntt_pass_kernel")[1].split("(")[0], g)
        out.setdefault(key, [[], []])[0 if s < last_r2c else 1].append((e - s) / 1e3)
    print(f"proof -{back}: upload ends {(last_r2c - r1[0][1]) / 1e6:.1f} ms into the round, transforms end {(r1[-1][2] - r1[0][1]) / 1e6:.1f} ms")
    for key, (a, b) in sorted(out.items()):
        if a and b:
            print(f"   {key[0]:28s} grid {key[1]:8d}: during upload {len(a):2d} x {sum(a) / len(a):6.1f} us   after {len(b):2d} x {sum(b) / len(b):6.1f} us   ratio {sum(a) / len(a) / (sum(b) / len(b)):.2f}")
