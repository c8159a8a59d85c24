#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into a per-kernel stats table (what
`rocprofv3 --kernel-trace --stats` summarises): calls, total/avg/min/max duration, % of GPU time.

    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, calls, tot, mn, mx, vg, sg, lds in rows:
        short = name.split("(")[0].replace("void oq::", "").replace("oq::", "")
        print(f"| {short} | {calls} | {tot/1e6:.3f} | {tot/calls/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} | {vg} | {sg} | {lds} |")
    print(f"\ntotal kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
