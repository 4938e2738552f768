#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace / stats run) as text:
per-kernel calls, total / average duration (us), share, VGPR / SGPR / LDS / scratch.
    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/xyz.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-70s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-70:]
        print("%-70s %6d %12.1f %12.2f %7.2f" % (short, calls, tot, avg, pct))
    print()
    print("%-50s %5s %5s %6s %8s %8s %14s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid x wg"))
    seen = set()
    q = ("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, grid_y, workgroup_x "
         "from kernels")
    for r in c.execute(q):
        short = r[0].replace("(anonymous namespace)::", "").split("(")[0][-50:]
        key = (short, r[6], r[7])
        if key in seen:
            continue
        seen.add(key)
        print("%-50s %5d %5d %6d %8d %8d %8dx%d/%d" % (short, r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
    try:
        rows = list(c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by 1,2"))
        if rows:
            print("\n%-50s %-28s %18s %8s" % ("kernel", "counter", "sum", "dispatches"))
            for name, cn, v, n in rows:
                short = name.replace("(anonymous namespace)::", "").split("(")[0][-50:]
                print("%-50s %-28s %18.1f %8d" % (short, cn, v, n))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
