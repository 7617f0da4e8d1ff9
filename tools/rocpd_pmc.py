#!/usr/bin/env python
"""Per-kernel PMC sums from a rocprofv3 rocpd sqlite database (one --pmc pass). Prints CSV: kernel,counter,dispatches,sum,avg.
usage: python tools/rocpd_pmc.py <trace.db> [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", (name or "").replace("(anonymous namespace)::", ""))
    return name.replace("void ", "").replace("r3m::", "")[:100]


def cols(db, t):
    return [c[1] for c in db.execute(f"pragma table_info('{t}')")]


def main():
    db = sqlite3.connect(sys.argv[1])
    out = []
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in views:
        c = cols(db, "counters_collection")
        out.append("# counters_collection columns: " + ",".join(c))
        kn = "kernel_name" if "kernel_name" in c else ("name" if "name" in c else None)
        cn = "counter_name" if "counter_name" in c else ("pmc_name" if "pmc_name" in c else None)
        cv = "value" if "value" in c else ("counter_value" if "counter_value" in c else None)
        did = "dispatch_id" if "dispatch_id" in c else ("id" if "id" in c else None)
        if kn and cn and cv:
            q = (f"select {kn}, {cn}, count(distinct {did}), sum({cv}) from counters_collection group by {kn}, {cn} "
                 f"order by sum({cv}) desc")
            out.append("kernel,counter,dispatches,sum,avg_per_dispatch")
            for k, n, d, s in db.execute(q):
                out.append(f"\"{short(k)}\",{n},{d},{s:.6g},{(s / d if d else 0):.6g}")
    else:
        out.append("# no counters_collection view; tables: " + ",".join(views))
        for t in ("pmc_events", "rocpd_pmc_event", "pmc_info", "rocpd_info_pmc"):
            if t in views:
                out.append(f"# {t} columns: " + ",".join(cols(db, t)))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
