#!/usr/bin/env python3
"""Per-kernel SQ counters of a rocprofv3 --pmc run (rocpd SQLite) beside the kernel durations of a --kernel-trace run of the same
command: VALU instructions per dispatch and SIMD cycles per VALU instruction (duration x 2.4 GHz x 1024 SIMDs / SQ_INSTS_VALU).
usage: sq_counters.py <pmc_results.db> <kernel_trace_results.db> [name filter]"""
import sqlite3, sys
pmc, kt = sqlite3.connect(sys.argv[1]), sqlite3.connect(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
def tables(db): return [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
dur = {}
for name, s, e in kt.execute("select name,start,end from kernels"):
    d = dur.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += (e - s)
t = tables(pmc)
view = "counters_collection" if "counters_collection" in t else next((x for x in t if "pmc" in x.lower() and "event" in x.lower()), None)
cols = [r[1] for r in pmc.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
ccol = "counter_name" if "counter_name" in cols else "counter"
vcol = "value" if "value" in cols else "counter_value"
dcol = "dispatch_id" if "dispatch_id" in cols else None
acc = {}
for row in pmc.execute(f"select {kcol},{ccol},sum({vcol}),count(distinct {dcol or kcol}) from {view} group by {kcol},{ccol}"):
    acc.setdefault(row[0], {})[row[1]] = (row[2], row[3])
print(f"{'kernel':66s} {'disp':>5s} {'VALU insts/disp':>16s} {'avg us':>9s} {'cycles/VALU':>12s} {'wave cycles/disp':>17s} {'wait inst any':>14s}")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", (0, 1))[0]):
    if flt and flt not in k: continue
    if "SQ_INSTS_VALU" not in c or k not in dur: continue
    insts, nd = c["SQ_INSTS_VALU"]
    per = insts / max(nd, 1)
    avg_ns = dur[k][1] / dur[k][0]
    cyc = avg_ns * 2.4 * 1024 / per if per else 0
    wc = c.get("SQ_WAVE_CYCLES", (0, 1)); wi = c.get("SQ_WAIT_INST_ANY", (0, 1))
    print(f"{k[:66]:66s} {nd:5d} {per:16.0f} {avg_ns / 1e3:9.1f} {cyc:12.2f} {wc[0] / max(wc[1], 1):17.0f} {wi[0] / max(wi[1], 1):14.0f}")
