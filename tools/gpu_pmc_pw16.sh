#!/bin/bash
# PMC passes on one conv_bench command for the persistent bf16 kernel. usage: gpu_pmc_pw16.sh <tag> <debug-bits> <conv_bench args...>   (probe build)
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); TAG=$1; DBG=$2; shift 2
export R3M_GG_DEBUG=$DBG
[ -n "$PROBE_LIB" ] && export R3M_HIP_LIB=$REPO/r3m_amd/lib/libr3m_hip_probes.so
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); d=/tmp/pmcp_${TAG}_$i; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $d -o pmc -- python $REPO/tools/conv_bench.py "$@" > /dev/null 2>&1)
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_pmc.py "$db" gpurun_out/pmcp_${TAG}_$i.csv > /dev/null
  rm -rf $d
done
cat gpurun_out/pmcp_${TAG}_*.csv | grep -i "pw16\|gather_gemm_bf16\|halo" | sort > gpurun_out/pmcp_${TAG}.txt
cat gpurun_out/pmcp_${TAG}.txt | cut -c1-30,90-200
