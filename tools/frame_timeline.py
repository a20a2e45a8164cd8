#!/usr/bin/env python
"""One frame of the incremental-mapping bench as the device saw it: every kernel between two launches of the frame's last
kernel (the importance sweep's fold), in order, with its duration and the idle gap in front of it; runs of a repeating
kernel pair (the 50 iterations) are folded into one line.  From a rocprofv3 --kernel-trace run (rocpd sqlite db).

    python tools/frame_timeline.py <rocprof dir> [anchor substring = k_sweep_fold] [frame = -2 (second to last)]
"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_sweep_fold"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
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
    if len(pos) < 3:
        sys.exit("fewer than 3 launches of %s" % anchor)
    a, b = pos[which - 1], pos[which]
    frame = rows[a + 1:b + 1]
    print("# source: %s" % dbs[0])
    print("# frame = launches %d..%d, %.3f ms from the end of the previous frame's last kernel to the end of this one's"
          % (a + 1, b, (rows[b][2] - rows[a][2]) / 1e6))
    print("# %-70s %6s %10s %10s" % ("kernel", "count", "dur_us", "gap_us"))
    i, prev_end = 0, rows[a][2]
    tot_d = tot_g = 0.0
    while i < len(frame):
        # a repeating pair (x, y, x, y, ...)
        if i + 3 < len(frame) and frame[i][0] == frame[i + 2][0] and frame[i + 1][0] == frame[i + 3][0] and frame[i][0] != frame[i + 1][0]:
            j = i
            while j + 1 < len(frame) and frame[j][0] == frame[i][0] and frame[j + 1][0] == frame[i + 1][0]:
                j += 2
            n = (j - i) // 2
            for k in (0, 1):
                d = sum(frame[m][2] - frame[m][1] for m in range(i + k, j, 2))
                g = sum(frame[m][1] - (frame[m - 1][2] if m > 0 else prev_end) for m in range(i + k, j, 2))
                tot_d, tot_g = tot_d + d, tot_g + g
                print("  %-70s %6d %10.2f %10.2f   (sums; per launch %.2f + %.2f)"
                      % (frame[i + k][0].replace(".kd", "")[:70], n, d / 1e3, g / 1e3, d / n / 1e3, g / n / 1e3))
            i = j
            continue
        r = frame[i]
        g = r[1] - (frame[i - 1][2] if i > 0 else prev_end)
        tot_d, tot_g = tot_d + (r[2] - r[1]), tot_g + g
        print("  %-70s %6d %10.2f %10.2f" % (r[0].replace(".kd", "")[:70], 1, (r[2] - r[1]) / 1e3, g / 1e3))
        i += 1
    print("# kernels %.3f ms + gaps %.3f ms, %d launches" % (tot_d / 1e6, tot_g / 1e6, len(frame)))


if __name__ == "__main__":
    main()
