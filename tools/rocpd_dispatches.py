#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains PATTERN, in time order: start (ms since the first listed one), duration,
grid, workgroup, LDS.   python tools/rocpd_dispatches.py results.db PATTERN [limit]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
gx = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "0")
rows = c.execute("select d.start, d.end, s.kernel_name, %s, %s, d.group_segment_size from rocpd_kernel_dispatch d "
                 "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like ? order by d.start" % (gx, wx),
                 ("%" + pat + "%",)).fetchall()
t0 = rows[0][0] if rows else 0
for st, en, name, g, w, lds in rows[:limit]:
    short = name.split("(")[0].replace("void oq::", "").replace("(anonymous namespace)::", "")
    print("%10.3f ms  %9.1f us  grid %9d  wg %4d  lds %6d  %s" % ((st - t0) / 1e6, (en - st) / 1e3, g, w, lds, short[:60]))
