#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from one or more rocpd sqlite databases
(counter instances of a dispatch are summed, then averaged over dispatches).
usage: rocprof_pmc_summary.py out.txt db1 [db2 ...]"""
import collections
import sqlite3
import sys

import numpy as np


def main():
    out = open(sys.argv[1], "w")
    out.write("# rocprofv3 --pmc per-kernel averages per dispatch (FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE\n"
              "# under-reports wide coalesced reads by 2x, see MI355X_MICROARCH.md #HBM — multiply by 2 before comparing)\n")
    for db in sys.argv[2:]:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value), avg(duration) from pmc_events "
                           "group by name, counter_name, dispatch_id").fetchall()
        agg = collections.defaultdict(list)
        for n, c, d, v, dur in rows:
            agg[(n, c)].append((v, dur))
        out.write("\n## %s\n" % db.split("/")[-2])
        for (n, c), vals in sorted(agg.items()):
            if len(vals) < 5:
                continue
            v = np.array([x[0] for x in vals])
            out.write("%-70s %-22s dispatches=%5d avg=%16.1f avg_kernel_us=%8.2f\n" %
                      (n[:70], c, len(vals), v.mean(), np.mean([x[1] for x in vals]) / 1e3))


if __name__ == "__main__":
    main()
