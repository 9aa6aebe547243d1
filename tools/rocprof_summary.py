#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd SQLite outputs (kernel trace + PMC passes) into a small text table for profiles/.
usage: rocprof_summary.py <kernel_trace.db> [<pmc.db> ...] [--bench-log <stdout of the profiled bench.py run>]"""
import json
import sqlite3
import sys


def bench_line(path):
    """The JSON line bench.py printed in the SAME (profiled) run: its avg_launch_ms is the box-local figure to hold against the
    kernel-duration sum below (different boxes of the pool differ by a few per cent; only same-run numbers compare)."""
    for line in open(path, errors="replace"):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def main():
    bench = None
    if "--bench-log" in sys.argv:
        i = sys.argv.index("--bench-log")
        bench = bench_line(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    kt = sqlite3.connect(sys.argv[1])
    print(f"# kernel trace: {sys.argv[1]}")
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in kt.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:90]:90s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
    passes = [(n, c, a) for n, c, t, a, p in kt.execute("select name,total_calls,total_duration,average,percentage from top_kernels") if "ntt_pass_kernel" in n]
    if passes:
        ntts = min(c for _, c, _ in passes)
        per_ntt = sum(c * a for _, c, a in passes) / ntts
        print(f"# NTT pass kernels: {sum(c for _, c, _ in passes)} launches = {ntts} transforms x {sum(c for _, c, _ in passes) / ntts:.0f} passes; "
              f"sum of the pass durations per transform {per_ntt:.1f} us")
        if bench:
            r = bench.get("roofline", {})
            print(f"# bench.py in this same run (under the profiler, same box): ms_per_step {bench.get('ms_per_step')}, roofline.avg_launch_ms "
                  f"{r.get('avg_launch_ms')} (HIP events around the whole chain, launch gaps included), value {bench.get('value'):.4g} {bench.get('unit')}")
    print()
    print("# per-kernel launch geometry (first dispatch of each kernel)")
    for row in kt.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels group by name"):
        print("  ", row)
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        print(f"\n# PMC pass: {path}")
        print(f"{'kernel':90s} {'counter':>12s} {'dispatches':>10s} {'avg_value':>14s}")
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for name, ctr, cnt, avg in db.execute(q):
            print(f"{name[:90]:90s} {ctr:>12s} {cnt:10d} {avg:14.1f}")


if __name__ == "__main__":
    main()
