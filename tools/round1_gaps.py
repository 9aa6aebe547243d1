#!/usr/bin/env python3
"""Round 1 of the LAST proof in a rocprofv3 kernel trace: busy time and gaps of the transform kernels up to the first leaf hash.
usage: round1_gaps.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "gather_jobs_kernel" in r[0]]
lo, hi = ends[-2] + 1, ends[-1] + 1
seg = rows[lo:hi]
first_leaf = next(i for i, r in enumerate(seg) if "leaf_hash" in r[0] and (r[2] - r[1]) > 2e6)
r1 = seg[:first_leaf]
ntt = [r for r in r1 if "ntt_pass" in r[0]]
t0, t1 = ntt[0][1], ntt[-1][2]
busy = sum(e - s for _, s, e in ntt)
# union of all kernel intervals in [t0, t1]
iv = sorted((s, e) for _, s, e in r1 if e > t0 and s < t1)
cover, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: cover += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cover += cur_e - cur_s
gaps = sorted(((ntt[i + 1][1] - ntt[i][2]) / 1e3 for i in range(len(ntt) - 1)), reverse=True)
print(f"round-1 transforms: {len(ntt)} launches, span {(t1 - t0) / 1e6:.2f} ms, sum of durations {busy / 1e6:.2f} ms, any kernel running {cover / 1e6:.2f} ms, "
      f"idle {(t1 - t0 - cover) / 1e6:.2f} ms; largest gaps between consecutive transform kernels (us): {[round(g) for g in gaps[:12]]}; gaps > 20 us: {sum(1 for g in gaps if g > 20)} totalling {sum(g for g in gaps if g > 20) / 1e3:.2f} ms")
by = {}
for n, s, e in ntt:
    k = n.split("(")[0][-40:]
    by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += e - s
for k, (c, d) in by.items():
    print(f"   {k}: {c} launches, {d / 1e6:.2f} ms, avg {d / c / 1e3:.1f} us")
