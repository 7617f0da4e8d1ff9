#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only) on ONE bench step; only compact CSV summaries are kept.
mkdir -p gpurun_out; export TMPDIR=/tmp
# usage: gpu_pmc.sh [clips] [fp32|bf16] [file suffix] [extra bench.py arguments, e.g. "--size 34 --doaug rctraj"]
REPO=$(pwd); CL=${1:-256}; PREC=${2:-fp32}; SFX=${3:-}; EXTRA=${4:-}
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1)); tag=pass$i$SFX
  rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$tag -o pmc -- python $REPO/bench.py --clips-per-gpu $CL --precision $PREC $EXTRA --steps 1 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > $REPO/gpurun_out/pmc_$tag.log 2>&1)
  echo "pmc $tag ($C) rc=$?"
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py "$db" gpurun_out/pmc_${tag}.csv > /dev/null; python tools/rocpd_stats.py "$db" gpurun_out/pmc_${tag}_kernels.csv > /dev/null; head -12 gpurun_out/pmc_${tag}.csv; fi
  tail -2 gpurun_out/pmc_$tag.log | cut -c1-200
  rm -rf /tmp/pmc_$tag
done
