#!/usr/bin/env python
"""Kernel sequence (start offset, duration, idle time before it) between the last launches of two named kernels of a rocprofv3
kernel trace (rocpd sqlite) — e.g. the objective between the encoder's forward and backward. usage: rocpd_window.py trace.db FROM TO"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
a, b = sys.argv[2], sys.argv[3]
ib = max(i for i, r in enumerate(rows) if b in (r[0] or ""))
ia = max(i for i, r in enumerate(rows[:ib]) if a in (r[0] or ""))
t0 = rows[ia][1]
prev_end = rows[ia - 1][2] if ia else rows[ia][1]
for name, s, e in rows[max(ia - 3, 0):ib + 4]:
    n = re.sub(r"\(.*$", "", name or "").replace("void ", "").replace("r3m::", "")[:70]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  idle-before {(s - prev_end) / 1e3:8.1f}  {n}")
    prev_end = max(prev_end, e)
