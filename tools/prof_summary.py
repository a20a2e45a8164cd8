#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (its rocpd sqlite db) as a per-kernel table.

    python tools/prof_summary.py gpurun_out/<dir> [top_n] > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not dbs:
        sys.exit("no *_results.db under %s" % root)
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "sum(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), "
         "max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 6 desc" % (kd, ks))
    rows = list(cur.execute(q))
    total = sum(r[5] for r in rows) or 1
    print("# source: %s" % dbs[0])
    print("# %-70s %7s %10s %10s %10s %7s %5s %5s %5s %7s %7s %9s %5s" % (
        "kernel", "calls", "avg_us", "min_us", "max_us", "%time", "vgpr", "agpr", "sgpr", "lds_B", "scr_B", "grid", "wg"))
    for r in rows[:top]:
        name = r[0].replace(".kd", "")
        print("  %-70s %7d %10.2f %10.2f %10.2f %6.1f%% %5s %5s %5s %7s %7s %9s %5s" % (
            name[:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / total, r[6], r[7], r[8], r[9], r[10],
            r[11], r[12]))


if __name__ == "__main__":
    main()
