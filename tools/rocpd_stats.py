#!/usr/bin/env python
"""Kernel summary (calls, total, average, %) from a rocprofv3 rocpd sqlite database — the same table `--stats` prints.
usage: python tools/rocpd_stats.py gpurun_out/prof/trace_results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("r3m::", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append(f"\"{short(name)}\",{calls},{tot/1e6:.3f},{avg/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100.0*tot/total:.2f}")
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    lines.append(f"# total kernel time {total/1e6:.3f} ms; first-start to last-end {(span[1]-span[0])/1e6:.3f} ms")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
