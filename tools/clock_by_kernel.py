#!/usr/bin/env python
"""Effective shader clock and matrix-pipe busy fraction per kernel from the third PMC pass of tools/gpu_pmc.sh (gpurun_out/pmc_pass3*.csv +
*_kernels.csv): usage clock_by_kernel.py [dir] > profiles/rNN_clock_by_kernel.txt"""
import csv, sys, collections
G = (sys.argv[1] if len(sys.argv) > 1 else "gpurun_out") + "/"
def table(sfx):
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(l for l in open(G + f"pmc_pass3{sfx}.csv") if not l.startswith("#")):
        cnt[r["kernel"]][r["counter"]] = float(r["sum"])
    ker = {r["kernel"]: (int(r["calls"]), float(r["total_ms"])) for r in csv.DictReader(open(G + f"pmc_pass3{sfx}_kernels.csv")) if r.get("total_ms")}
    rows = []
    for k, c in cnt.items():
        if k not in ker or "GRBM_GUI_ACTIVE" not in c: continue
        n, ms = ker[k]
        gui = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
        ghz = gui / (ms * 1e-3) / 1e9
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024) if gui else 0.0
        rows.append((ms, k, n, ghz, busy))
    rows.sort(reverse=True)
    return rows
for sfx, name in (("", "fp32 headline step"), ("_bf16", "bf16 ResNet-50 step"), ("_bf16_r34", "bf16 ResNet-34 rctraj step (configs[4])")):
    print(f"# {name}: effective shader clock and matrix-pipe busy fraction per kernel (one bench step under rocprofv3 --pmc")
    print("# 'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES'; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; busy = MFMA busy cycles / (GUI / 8 x 1024 SIMDs);")
    print("# kernels of a few microseconds read too high: the counter window is wider than the kernel)")
    print(f"{'kernel':72s} {'launches':>8s} {'ms':>8s} {'GHz':>6s} {'mfma busy':>9s}")
    for ms, k, n, ghz, busy in table(sfx)[:28]:
        print(f"{k[:72]:72s} {n:8d} {ms:8.2f} {ghz:6.2f} {busy:9.3f}")
    print()
