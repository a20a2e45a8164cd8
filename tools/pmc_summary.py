#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 --pmc run (rocpd sqlite db) -> small text table.

    python tools/pmc_summary.py <dir> [kernel-substring]
"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not dbs:
        sys.exit("no *_results.db under %s" % root)
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    if not pmc or not info:
        print("tables:", tabs)
        sys.exit("no pmc tables")
    print("# source:", dbs[0])
    cols_e = [c[1] for c in cur.execute("pragma table_info(%s)" % pmc[0])]
    cols_i = [c[1] for c in cur.execute("pragma table_info(%s)" % info[0])]
    print("# pmc_event cols:", cols_e)
    print("# info_pmc cols:", cols_i)
    q = ("select s.kernel_name, i.name, count(*), sum(e.value), avg(e.value) from %s e join %s i on e.pmc_id = i.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, i.name "
         "order by s.kernel_name" % (pmc[0], info[0], kd, ks))
    try:
        for r in cur.execute(q):
            if pat in r[0]:
                print("%-60s %-22s n=%6d sum=%16.1f avg=%14.2f" % (r[0][:60], r[1], r[2], r[3], r[4]))
    except Exception as e:
        print("query failed:", e)


if __name__ == "__main__":
    main()
