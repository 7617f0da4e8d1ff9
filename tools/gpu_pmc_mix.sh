#!/bin/bash
# Instruction mix (per MFMA) of every GEMM kernel a conv_bench command launches: one rocprofv3 --pmc pass. usage: gpu_pmc_mix.sh <tag> <conv_bench args...>
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); TAG=$1; shift
d=/tmp/pmcx_$TAG; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $d -o pmc -- python $REPO/tools/conv_bench.py "$@" > /dev/null 2>&1)
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_pmc.py "$db" gpurun_out/pmcx_$TAG.csv > /dev/null
python - <<PY
import csv, collections
k = collections.defaultdict(dict)
for r in csv.reader(open("gpurun_out/pmcx_$TAG.csv")):
    if len(r) >= 5 and r[1].startswith("SQ_"):
        k[r[0]][r[1]] = float(r[4])
for name, c in k.items():
    m = c.get("SQ_INSTS_MFMA", 0)
    if m <= 0: continue
    print(f"{name[:64]:64s} per MFMA: valu {(c['SQ_INSTS_VALU'] - m) / m:5.2f} salu {c['SQ_INSTS_SALU'] / m:5.2f} lds {c['SQ_INSTS_LDS'] / m:5.2f} vmem {c['SQ_INSTS_VMEM_RD'] / m:5.2f}  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c['SQ_BUSY_CYCLES'], 1) / 4:5.3f}")
PY
