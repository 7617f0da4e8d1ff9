#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only) on ONE bench step; rocpd databases land in gpurun_out/pmc_*
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); CL=${1:-256}
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $REPO/gpurun_out/pmc_$tag -o pmc -- python $REPO/bench.py --clips-per-gpu $CL --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
cd $REPO; du -sh gpurun_out/pmc_* | head
