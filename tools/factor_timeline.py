#!/usr/bin/env python3
"""Busy / idle time of the device inside the numeric factorisations of a rocprofv3 kernel trace (rocpd sqlite): every span from a
k_diag_init to the next k_gather_csr (level-scheduled solves) or k_sn_invert (supernodal solves).   python tools/factor_timeline.py results.db"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                 "on d.kernel_id = s.id order by d.start").fetchall()
inside, t0, busy, prev_end, n = False, 0, 0.0, 0, 0
by = defaultdict(lambda: [0, 0.0, 0.0])
for st, en, name in rows:
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    k = m.group(1) if m else name[:30]
    if "k_diag_init" in name:
        inside, t0, busy, prev_end, n = True, st, 0.0, st, 0
        by.clear()
    if inside:
        e = by[k]
        e[0] += 1; e[1] += (en - st) / 1e3; e[2] += max(0.0, (st - prev_end) / 1e3)
        busy += (en - st) / 1e3
        prev_end = max(prev_end, en)
        n += 1
        # the last kernel of a factorisation: the CSR copy of the values (level-scheduled solves), the block inversion
        # (supernodal solves, level-by-level factorisation) or the gather of the supernodes' lists (supernodal solves after a
        # multifrontal factorisation: the fronts invert their blocks themselves)
        if "k_gather_csr" in name or "k_sn_invert" in name or ("k_sn_gather" in name and any("k_mf_front" in q for q in by)):
            span = (en - t0) / 1e3
            print("factorisation: %d launches, span %.1f ms, kernels %.1f ms, idle %.1f ms" % (n, span / 1e3, busy / 1e3, (span - busy) / 1e3))
            for kk, (cnt, dur, idle) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:8]:
                print("   %-22s %5d launches  %8.2f ms busy  %8.2f ms idle before (%.1f us each)" % (kk, cnt, dur / 1e3, idle / 1e3, idle / cnt))
            inside = False
