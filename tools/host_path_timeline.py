#!/usr/bin/env python3
"""Timeline of the last host-buffer proof in a rocprofv3 trace of tools/prove_bench.py (--kernel-trace --memory-copy-trace):
the H2D copies of the upload pipeline against the kernels of round 1.  usage: host_path_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if "memory_cop" in t.lower()]
print("memory copy tables:", mc[:6])
kern = list(db.execute("select name,start,end from kernels order by start"))
copies = []
for t in mc:
    cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
    if "start" in cols and "end" in cols:
        size_col = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
        name_col = "name" if "name" in cols else None
        q = f"select {name_col or 'NULL'}, start, end, {size_col or '0'} from {t} order by start"
        copies = list(db.execute(q))
        break
big = [c for c in copies if c[3] and c[3] > (32 << 20)]
print(f"{len(kern)} kernels, {len(copies)} copies, {len(big)} copies > 32 MB")
if not big:
    sys.exit(0)
# the last proof: the last run of consecutive big copies
last = [big[-1]]
for c in reversed(big[:-1]):
    if last[0][1] - c[2] < 50e6:
        last.insert(0, c)
    else:
        break
t0 = last[0][1]
print(f"last upload: {len(last)} column groups")
for c in last:
    print(f"  copy {c[3] / 1e6:8.1f} MB  start {(c[1] - t0) / 1e6:7.2f} ms  end {(c[2] - t0) / 1e6:7.2f} ms  ({c[3] / (c[2] - c[1]):.1f} GB/s)")
tend = last[-1][2] + 30e6
ks = [k for k in kern if t0 - 2e6 <= k[1] <= tend]
busy, prev_end, gaps = 0, None, []
for name, s, e in ks:
    busy += e - s
    if prev_end is not None and s - prev_end > 200e3:
        gaps.append((prev_end - t0, s - prev_end, name))
    prev_end = max(prev_end or e, e)
print(f"kernels in the window: {len(ks)}, busy {busy / 1e6:.2f} ms, first kernel at {(ks[0][1] - t0) / 1e6:.2f} ms")
for at, g, nm in gaps:
    print(f"  idle {g / 1e6:6.2f} ms at {at / 1e6:7.2f} ms before {nm[:60]}")
