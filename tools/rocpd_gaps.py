#!/usr/bin/env python
"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd sqlite): where does the device wait for the host?
Prints the idle total, a histogram of gap lengths and the largest gaps with the kernels on either side.
usage: python tools/rocpd_gaps.py trace.db [skip_first_ms]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name or "")
    return name.replace("void ", "").replace("r3m::", "")[:60]


db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
t0 = rows[0][1]
rows = [r for r in rows if r[1] - t0 >= skip]
busy = sum(r[2] - r[1] for r in rows)
span = rows[-1][2] - rows[0][1]
gaps = []
cur_end = rows[0][2]
for i in range(1, len(rows)):
    g = rows[i][1] - cur_end
    if g > 0:
        gaps.append((g, short(rows[i - 1][0]), short(rows[i][0]), (rows[i][1] - rows[0][1]) / 1e6))
    cur_end = max(cur_end, rows[i][2])
idle = sum(g[0] for g in gaps)
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy(sum) {busy/1e6:.2f} ms  idle(gaps) {idle/1e6:.2f} ms = {100*idle/span:.1f} %")
edges = [0, 2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e6, 1e12]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print(f"  gaps {lo/1e3:7.0f}..{hi/1e3:9.0f} us: n={len(sel):5d} total {sum(sel)/1e6:8.3f} ms")
print("largest gaps (us, after kernel -> before kernel, at ms):")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]/1e3:9.1f}  {g[1]} -> {g[2]}  @{g[3]:.1f}")
# per (prev kernel) aggregate of small gaps
agg = {}
for g in gaps:
    a = agg.setdefault(g[1], [0, 0.0])
    a[0] += 1
    a[1] += g[0]
print("idle by preceding kernel:")
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {s/1e6:8.3f} ms  n={n:5d}  avg {s/n/1e3:7.1f} us  after {k}")
