#!/usr/bin/env python
"""Merge the per-pass PMC CSVs written by tools/gpu_pmc.sh into one per-kernel table + profiles/pmc_latest.json.
Corrections follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide (16 B/lane) coalesced reads -> doubled here (our kernels read with dwordx4). GRBM_GUI_ACTIVE is summed over
the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs.
usage: pmc_report.py <dir with pmc_pass*.csv> <out_prefix> [pass-file suffix] [json name] [resnet size] [clips per GPU]
(the last two = the workload the passes ran, default 50 / 256 = tools/gpu_pmc.sh's; bench.py refuses a summary of another workload)"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def csrc_sha16():
    """The build these counters were collected on: sha256 over r3m_amd/csrc/*.{hip,h} (sorted by name), first 16 hex digits — the
    same function as bench.py's, which says in its line whether `roofline.traffic` comes from the build it is timing."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "r3m_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "r3m_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


d, outp = sys.argv[1], sys.argv[2]
sfx = sys.argv[3] if len(sys.argv) > 3 else ""
json_name = sys.argv[4] if len(sys.argv) > 4 else "pmc_latest.json"
wl_size = int(sys.argv[5]) if len(sys.argv) > 5 else 50
wl_clips = int(sys.argv[6]) if len(sys.argv) > 6 else 256
cnt = collections.defaultdict(dict)      # kernel -> counter -> (dispatches, sum)
dur = {}
for i in range(1, 5):
    try:
        for row in csv.reader(l for l in open(f"{d}/pmc_pass{i}{sfx}.csv") if not l.startswith("#")):
            if row[0] == "kernel":
                continue
            cnt[row[0]][row[1]] = (int(row[2]), float(row[3]))
    except FileNotFoundError:
        pass
    try:
        for row in csv.reader(l for l in open(f"{d}/pmc_pass{i}{sfx}_kernels.csv") if not l.startswith("#")):
            if row[0] == "kernel":
                continue
            dur.setdefault(row[0], (int(row[1]), float(row[3])))   # calls, avg_us (first pass seen)
    except FileNotFoundError:
        pass


def group(k):
    k = short_name(k)
    k = re.sub(r"gather_gemm(_glds2?|_bf16)?_kernel<128, 128, 2, 2, \d+(, \d+)*>", "gather_gemm 128x128 (all epilogues)", k)
    k = re.sub(r"gather_gemm(_glds2?|_bf16)?_kernel<256, 64, 4, 1, \d+(, \d+)*>", "gather_gemm 256x64 (all epilogues)", k)
    k = re.sub(r"gather_gemm_k16_kernel<\d+>", "gather_gemm 128x128 (all epilogues)", k)      # 16-wide-K variant of the same tile (round 2)
    k = re.sub(r"pw_gemm_kernel<128, 128, 2, 2, \d+(, \w+)*>", "gather_gemm 128x128 (all epilogues)", k)   # persistent 1x1 kernel (round 4): same launch class
    # the window / halo kernels' launches belong to the same launch classes as the gather kernel's (bench.py's roofline classes)
    k = re.sub(r"conv3x3_win_kernel<128, 128, 2, 2, \d+, \d+>", "gather_gemm 128x128 (all epilogues)", k)
    k = re.sub(r"conv3x3_halo_bf16_kernel<(128|256), 128, 2, 2, \d+>", "gather_gemm 128x128 (all epilogues)", k)
    k = re.sub(r"conv3x3_halo_bf16_kernel<256, 64, 4, 1, \d+>", "gather_gemm 256x64 (all epilogues)", k)
    # round 6: the persistent kernel-row 3x3 kernel (conv_row16.hip): <BN, EPI>
    k = re.sub(r"conv3x3_row_bf16_kernel<128, \d+>", "gather_gemm 128x128 (all epilogues)", k)
    k = re.sub(r"conv3x3_row_bf16_kernel<64, \d+>", "gather_gemm 256x64 (all epilogues)", k)
    # round 5: the persistent big-tile bf16 kernel (conv_pw16.hip): wide outputs / 64-channel outputs
    k = re.sub(r"pw16_gemm_kernel<256, (128|256), \d+, \d+(, \d+)*>", "gather_gemm 128x128 (all epilogues)", k)
    k = re.sub(r"pw16_gemm_kernel<(256|512), 64, \d+, \d+(, \d+)*>", "gather_gemm 256x64 (all epilogues)", k)
    return k


dur_norm = {short_name(k): v for k, v in dur.items()}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for k, cs in cnt.items():
    gk = group(k)
    for c, (n, s) in cs.items():
        agg[gk][c] += s
        agg[gk]["n_" + c] += n
    # the counter CSV keeps mangled names for kernels with bf16 arguments (the tracer's demangler does not know DF16b) while the
    # duration CSV (tools/rocpd_stats.py) already shortened them: join on the normalised name (round 1 looked up the raw key and
    # every bn_*16 row came out with avg_us = 0)
    d = dur.get(k) or dur_norm.get(short_name(k))
    if d:
        agg[gk]["calls"] += d[0]
        agg[gk]["time_us"] += d[0] * d[1]

rows = []
for k, a in agg.items():
    n = a.get("n_FETCH_SIZE") or a.get("calls") or 1
    t_us = a["time_us"] / max(a["calls"], 1)
    rd = a.get("FETCH_SIZE", 0.0) * 1024 * 2 / n
    wr = a.get("WRITE_SIZE", 0.0) * 1024 / max(a.get("n_WRITE_SIZE", n), 1)
    gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mfma_util = mf / (gui * 1024) if gui else 0.0
    ldsa, ldsc = a.get("SQ_LDS_IDX_ACTIVE", 0.0), a.get("SQ_LDS_BANK_CONFLICT", 0.0)
    wc, wi = a.get("SQ_WAVE_CYCLES", 0.0), a.get("SQ_WAIT_INST_ANY", 0.0)
    rows.append(dict(kernel=k, launches=int(n), avg_us=round(t_us, 1), hbm_read_MB_per_launch=round(rd / 1e6, 1),
                     hbm_write_MB_per_launch=round(wr / 1e6, 1),
                     hbm_GBps=round((rd + wr) / (t_us * 1e-6) / 1e9, 1) if t_us else 0.0,
                     mfma_util=round(mfma_util, 4), lds_conflict_frac=round(ldsc / ldsa, 4) if ldsa else 0.0,
                     wait_inst_frac=round(wi / wc, 4) if wc else 0.0, total_ms=round(a["time_us"] / 1e3, 2)))
rows.sort(key=lambda r: -r["total_ms"])
keys = list(rows[0].keys())
with open(outp + ".csv", "w") as f:
    f.write(",".join(keys) + "\n")
    for r in rows:
        f.write(",".join(f"\"{r[k]}\"" if k == "kernel" else str(r[k]) for k in keys) + "\n")
dom = rows[0]
json.dump({"source": outp + ".csv", "workload": {"size": wl_size, "clips": wl_clips}, "dominant_kernel": dom["kernel"],
           "dominant_kernel_hbm_bytes_per_launch": int((dom["hbm_read_MB_per_launch"] + dom["hbm_write_MB_per_launch"]) * 1e6),
           "dominant_kernel_mfma_util": dom["mfma_util"], "note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB units",
           "csrc_sha16": csrc_sha16()},
          open("profiles/" + json_name, "w"), indent=1)
for r in rows[:16]:
    print(r)
