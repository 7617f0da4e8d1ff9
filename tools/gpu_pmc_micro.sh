#!/bin/bash
# PMC passes on ONE micro-benchmark command (tools/conv_bench.py ...): usage gpu_pmc_micro.sh <tag> <conv_bench args...>
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); TAG=$1; shift
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); d=/tmp/pmcm_${TAG}_$i; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $d -o pmc -- python $REPO/tools/conv_bench.py "$@" > /dev/null 2>&1)
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_pmc.py "$db" gpurun_out/pmcm_${TAG}_$i.csv > /dev/null
  rm -rf $d
done
cat gpurun_out/pmcm_${TAG}_*.csv | grep -i "wgrad\|gather\|halo" | grep -v reduce | sort | head -60
