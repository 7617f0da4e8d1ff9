#!/usr/bin/env python
"""Copy the summaries of the last tools/gpu_evidence.sh run from gpurun_out/ (scratch) into profiles/ (tracked) under one tag:
usage: collect_profiles.py r04_v1"""
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
G, P = "gpurun_out", "profiles"
copy = {"bench_fp32.json": "bench_fp32.json", "bench_bf16.json": "bench_bf16.json", "kt_fp32_bench.json": "bench_fp32_traced_run.json",
        "kt_bf16_bench.json": "bench_bf16_traced_run.json", "kt_r34c4_bench.json": "bench_c4_traced_run.json",
        "kernel_stats_fp32.csv": "kernel_stats_fp32.csv", "kernel_stats_bf16.csv": "kernel_stats_bf16.csv",
        "kernel_stats_r34c4.csv": "kernel_stats_c4_r34_bf16_rctraj.csv", "parity.txt": "parity_report.txt", "gpu.txt": "gpu.txt"}
for f in os.listdir(G):
    if f.startswith("cfg_") and f.endswith(".json"):
        copy[f] = f
for src, dst in copy.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}"))
for prec, sfx, js, size, clips in (("fp32", "", "pmc_latest.json", 50, 256), ("bf16", "_bf16", "pmc_latest_bf16.json", 50, 256),
                                   ("bf16_r34", "_bf16_r34", "pmc_latest_bf16_r34.json", 34, 512)):
    if os.path.exists(os.path.join(G, f"pmc_pass1{sfx}.csv")):
        subprocess.check_call([sys.executable, "tools/pmc_report.py", G, os.path.join(P, f"{tag}_pmc_{prec}"), sfx, js, str(size), str(clips)])
for prec in ("fp32", "bf16"):
    csvf = os.path.join(G, f"launches_{prec}.csv")
    if os.path.exists(csvf):
        out = subprocess.run([sys.executable, "tools/launch_report.py", csvf, "20"], capture_output=True, text=True).stdout
        open(os.path.join(P, f"{tag}_launch_report_{prec}.txt"), "w").write(out)
        out = subprocess.run([sys.executable, "tools/launch_report.py", csvf, "20", "--roofs", prec], capture_output=True, text=True).stdout
        open(os.path.join(P, f"{tag}_tworoof_{prec}.txt"), "w").write(out)
for prec in ("fp32", "bf16"):
    src = os.path.join(G, f"power_{prec}.txt")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_power_{prec}.txt"))
log = os.path.join(G, "pytest_all.log")
if os.path.exists(log):
    lines = [l for l in open(log, errors="replace") if l.startswith(("PASSED", "FAILED", "ERROR")) or " passed" in l or " failed" in l]
    open(os.path.join(P, f"{tag}_gpu_tests.txt"), "w").writelines(lines)
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))
