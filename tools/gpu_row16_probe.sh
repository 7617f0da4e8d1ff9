#!/bin/bash
# Timing probes of the round-6 bf16 3x3 kernel (csrc/conv_row16.hip) on the probe build: R3M_GG_DEBUG bits 1 no epilogue, 2 no DMA in
# the loop, 4 no fragment reads, 8 no vmcnt wait at the step barrier, 16 no barrier. usage: gpu_row16_probe.sh [out]
OUT=${1:-gpurun_out/row16_probe.txt}
mkdir -p gpurun_out
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so ONLY_NEW=1 REPS=1
for data in ${DATAS:-zeros randn}; do
  for dbg in ${PROBES:-0 1 2 3 4 7 16 31}; do
    echo "== data $data probe $dbg"
    DATA=$data R3M_GG_DEBUG=$dbg timeout 120 python tools/row16_check.py bench 2>&1 | grep fwd16
  done
done | tee $OUT
