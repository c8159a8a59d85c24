#!/usr/bin/env python3
"""Per-kernel averages of PMC counters from a rocprofv3 rocpd database.
    python tools/rocpd_pmc.py <results.db> [kernel-substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute(
    "select s.kernel_name, p.name, count(*), avg(e.value), avg(d.end - d.start) "
    "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
    "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
    "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name").fetchall()
for name, pmc, n, avg, dur in rows:
    if flt in name:
        print(f"{name.split('(')[0][:60]:60s} {pmc:28s} n={n:4d} avg={avg:.4g} avg_dur_us={dur/1e3:.1f}")
