#!/usr/bin/env python
"""Per-shape report from bench.py --launch-csv (one row per conv GEMM launch): usage launch_report.py file.csv steps"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    k = (int(r["class"]), int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]))
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
print("cls         M     N     K taps n/step ms/launch   TF/s  ms/step")
tot = collections.defaultdict(float)
for k, (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]} {k[1]:11d} {k[2]:5d} {k[3]:5d} {k[4]:3d} {n/steps:6.1f} {ms/n:9.3f} {gf/ms:7.1f} {ms/steps:8.2f}")
    tot[k[0]] += ms / steps
print({k: round(v, 2) for k, v in tot.items()}, round(sum(tot.values()), 2))
