#!/usr/bin/env python3
"""Dispatch timeline of the last `count` kernels of a rocprofv3 (rocpd sqlite) kernel trace: start (us, relative), duration,
idle time of the device before the kernel, short kernel name; then the idle time per kernel name over the window.

    python tools/rocpd_timeline.py results.db [count] > profiles/r03_rand1e5_timeline.md
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main(path, count=400):
    c = sqlite3.connect(path)
    rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[-count:]
    t0 = rows[0][0]
    prev_end = rows[0][0]
    idle_by = defaultdict(lambda: [0, 0.0, 0.0])
    print("```")
    for st, en, name in rows:
        idle = max(0.0, (st - prev_end) / 1e3)
        k = short(name)
        print("%9.1f %7.1f %6.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, idle, k))
        e = idle_by[k]
        e[0] += 1; e[1] += idle; e[2] += (en - st) / 1e3
        prev_end = max(prev_end, en)
    print("```")
    span = (rows[-1][1] - t0) / 1e3
    busy = sum(v[2] for v in idle_by.values())
    print("\nwindow %.1f us, kernels %.1f us, idle %.1f us (%.1f %%)\n" % (span, busy, span - busy, 100 * (span - busy) / span))
    print("| kernel | dispatches | mean duration us | mean idle before us |")
    print("|---|---|---|---|")
    for k, (n, idle, dur) in sorted(idle_by.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.2f | %.2f |" % (k, n, dur / n, idle / n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
