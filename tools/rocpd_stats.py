#!/usr/bin/env python
"""Kernel summary (calls, total, average, %) from a rocprofv3 rocpd sqlite database — the same table `--stats` prints.
usage: python tools/rocpd_stats.py gpurun_out/prof/trace_results.db [out.csv]"""
import re
import sqlite3
import sys

def short_name(k):
    """kernel-trace names of kernels with bf16 arguments come back mangled (the tracer's demangler does not know DF16b): keep
    namespace-less function name + template arguments in their mangled form"""
    k = k.replace("_ZN3r3m12_GLOBAL__N_1", "_ZN3r3m")          # kernels of an anonymous namespace inside r3m (conv_row16.hip)
    m = re.match(r"_ZN3r3m(\d+)", k)
    if not m:
        return k
    n = int(m.group(1))
    name = k[m.end():m.end() + n]
    rest = k[m.end() + n:]
    t = re.match(r"I(.*?)EEv", rest)
    if t:
        args = re.findall(r"Li(\d+)E|(DF16b)|(f)", t.group(1))
        name += "<" + ", ".join(a or ("bf16" if b else "float") for a, b, c in args) + ">"
    return name



def short(name):
    name = short_name(name).replace("(anonymous namespace)::", "")
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
