#!/usr/bin/env python3
"""Every kernel of a rocprofv3 kernel trace (rocpd SQLite) that runs longer than a threshold or follows a gap longer than one:
offset, duration, gap, grid, name.  usage: rocprof_all.py <kernel_trace.db> [min_dur_us=300] [min_gap_us=1000]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
min_dur = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
t0, prev = rows[0][1], rows[0][1]
print(f"# {len(rows)} kernels; shown: longer than {min_dur} us or behind a gap longer than {min_gap} us")
for i, (name, s, e, grid) in enumerate(rows):
    if (e - s) / 1e3 >= min_dur or (s - prev) / 1e3 >= min_gap:
        print(f"{i:6d} {(s - t0) / 1e3:11.1f} {(e - s) / 1e3:9.1f} {(s - prev) / 1e3:9.1f} {grid:10d}  {name.split('(')[0][-50:]}")
    prev = max(prev, e)
