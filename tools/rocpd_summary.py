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
    print("# per launch shape (the B = 1 latency launches of bench.py are separate rows from the timed batch launches)")
    print("%-50s %14s %6s %12s %12s %5s %5s %6s %8s %8s" % ("kernel", "grid x wg", "calls", "total_us", "avg_us", "vgpr", "agpr", "sgpr", "lds", "scratch"))
    q = ("select name, grid_x, grid_y, workgroup_x, count(*), sum(duration), avg(duration), max(vgpr_count), "
         "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels "
         "group by name, grid_x, grid_y, workgroup_x order by sum(duration) desc")
    for r in c.execute(q):
        short = r[0].replace("(anonymous namespace)::", "").split("(")[0][-50:]
        print("%-50s %14s %6d %12.1f %12.2f %5d %5d %6d %8d %8d" % (short, "%dx%d/%d" % (r[1], r[2], r[3]), r[4], r[5] / 1e3,
                                                                       r[6] / 1e3, r[7], r[8], r[9], r[10], r[11]))
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
