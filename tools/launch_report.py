#!/usr/bin/env python
"""Per-shape report from bench.py --launch-csv (one row per conv GEMM launch).

    launch_report.py file.csv steps                 one line per (class, M, N, K, taps): launches, ms, TFLOP/s
    launch_report.py file.csv steps --roofs fp32    launches split by their algorithmic bytes too (= by epilogue: a dgrad that also
                                                    reads the residual gradient, the consumer BatchNorm's y and mask words moves 3x
                                                    the bytes of the plain launch of the same shape) and priced against BOTH roofs:
                                                    t_mfma = FLOP / dense MFMA peak of the dtype, t_hbm = algorithmic bytes / 6.3 TB/s
                                                    (the achievable HBM rate), bound = max of the two,
                                                    eff = bound / measured. The last line sums bound and measured over the step.
Peaks: MI355X_MICROARCH.md (fp32 MFMA 157.3 TFLOP/s, bf16 2500 TFLOP/s dense, HBM3E 8 TB/s spec)."""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
roofs = None
if "--roofs" in sys.argv:
    roofs = sys.argv[sys.argv.index("--roofs") + 1]
    args = [a for a in args if a != roofs]
rows = list(csv.DictReader(open(args[0])))
steps = int(args[1]) if len(args) > 1 else 1
CLS = {0: "fwd/dgrad 128-wide", 1: "fwd/dgrad 64-wide", 2: "wgrad 128-wide", 3: "wgrad 64-wide"}

if roofs is None:
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
    sys.exit(0)

PEAK_TF = {"fp32": 157.3, "bf16": 2500.0}[roofs]
HBM_ACH, HBM_SPEC = 6300.0, 8000.0          # GB/s
agg = collections.OrderedDict()
for r in rows:
    k = (int(r["class"]), int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]), round(float(r["alg_mbytes"])))
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"]); a[3] += float(r["alg_mbytes"])
print(f"# two-roof view, {roofs}: t_mfma at {PEAK_TF} TFLOP/s, t_hbm at {HBM_ACH / 1000} TB/s (achievable; spec {HBM_SPEC / 1000}); bound = max; eff = bound / measured")
print("cls         M     N     K taps  alg MB n/step ms/launch   TF/s   GB/s  t_mfma  t_hbm  bound-by  eff   ms/step  bound/step")
tot_ms = collections.defaultdict(float)
tot_bd = collections.defaultdict(float)
for k, (n, ms, gf, mb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    t_mfma = gf / n / PEAK_TF            # GFLOP / (TFLOP/s) = ms
    t_hbm = mb / n / HBM_ACH             # MB / (GB/s) = ms
    bound = max(t_mfma, t_hbm)
    by = "mfma" if t_mfma >= t_hbm else "hbm"
    print(f"{k[0]} {k[1]:11d} {k[2]:5d} {k[3]:5d} {k[4]:3d} {k[5]:8d} {n/steps:6.1f} {per:9.3f} {gf/ms:7.1f} {mb/ms:6.0f} {t_mfma:7.3f} {t_hbm:6.3f}  {by:>6s}  {bound/per:5.2f} {ms/steps:9.2f} {bound*n/steps:9.2f}")
    tot_ms[k[0]] += ms / steps
    tot_bd[k[0]] += bound * n / steps
for c in sorted(tot_ms):
    print(f"# class {c} ({CLS.get(c, '?')}): measured {tot_ms[c]:.2f} ms/step, two-roof bound {tot_bd[c]:.2f} ms/step -> {tot_bd[c] / tot_ms[c]:.3f}")
print(f"# all conv GEMM launches: measured {sum(tot_ms.values()):.2f} ms/step, two-roof bound {sum(tot_bd.values()):.2f} ms/step -> {sum(tot_bd.values()) / sum(tot_ms.values()):.3f}")
