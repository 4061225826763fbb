#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the per-kernel table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration, share of GPU time.
usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(grid_x), max(workgroup_x), max(lds_size), max(scratch_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    out.write("# rocprofv3 --kernel-trace summary of %s\n" % db)
    out.write("%-96s %8s %12s %10s %10s %10s %6s %10s %5s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "wg", "lds"))
    for r in rows:
        out.write("%-96s %8d %12.1f %10.2f %10.2f %10.2f %6.2f %10d %5d %6d\n" %
                  (r[0][:96], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8]))
    try:
        pm = cur.execute("select name, count(*), avg(value), sum(value) from pmc_events group by name").fetchall()
        if pm:
            out.write("\n# counters (per dispatch average, total)\n")
            for r in pm:
                out.write("%-40s n=%8d avg=%16.2f total=%18.1f\n" % r)
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main()
