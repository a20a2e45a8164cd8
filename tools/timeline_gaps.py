#!/usr/bin/env python
"""Where a step's wall time goes BETWEEN kernels: from a rocprofv3 --kernel-trace run (rocpd sqlite db), the launches of
the fused step kernel delimit steps; steps are grouped by their launch sequence (the timed loop, the eager warm-up and the
full-iteration leg differ) and for every group seen >= min_count times: per position the kernel, its mean duration and the
mean idle gap in front of it.

    python tools/timeline_gaps.py <rocprof dir> [anchor kernel substring = k_step_v3] [min_count = 20]
"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_step_v3"
    min_count = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not dbs:
        sys.exit("no *_results.db under %s" % root)
    cur = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                            % (kd, ks)))
    pos = [i for i, r in enumerate(rows) if anchor in r[0]]
    groups = {}  # launch sequence between two anchors -> aggregate
    for a, b in zip(pos[:-1], pos[1:]):
        sig = tuple(rows[i][0] for i in range(a + 1, b + 1))
        g = groups.setdefault(sig, {"n": 0, "wall": 0.0, "dur": [0.0] * len(sig), "gap": [0.0] * len(sig)})
        g["n"] += 1
        g["wall"] += rows[b][1] - rows[a][1]
        for j in range(len(sig)):
            r, prev = rows[a + 1 + j], rows[a + j]
            g["dur"][j] += r[2] - r[1]
            g["gap"][j] += r[1] - prev[2]
    print("# source: %s" % dbs[0])
    for sig, g in sorted(groups.items(), key=lambda kv: -kv[1]["n"]):
        n = g["n"]
        if n < min_count:
            continue
        print("# %d steps of %d launches, mean step period %.2f us (anchor start to anchor start)" % (n, len(sig), g["wall"] / n / 1e3))
        print("# %-3s %-72s %10s %12s" % ("pos", "kernel (in launch order after the anchor)", "avg_us", "gap_before_us"))
        for j, name in enumerate(sig):
            print("  %-3d %-72s %10.2f %12.2f" % (j, name.replace(".kd", "")[:72], g["dur"][j] / n / 1e3, g["gap"][j] / n / 1e3))
        print("# kernels %.2f us + gaps %.2f us" % (sum(g["dur"]) / n / 1e3, sum(g["gap"]) / n / 1e3))

if __name__ == "__main__":
    main()
