#!/bin/bash
# probe: 3-stage ring with 32-wide K tiles (3 blocks/CU) against the shipped 2-stage form (4 blocks/CU), bf16 1x1 shapes of ResNet-50
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
S="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,7,512,2048,1,1,0 1280,28,256,512,1,1,0 1280,14,512,1024,1,1,0 1280,56,256,128,1,1,0 1280,28,512,256,1,1,0"
for rep in 1 2; do
  echo "== 2 stages"; R3M_BF16_NST3=0 python tools/conv_bench.py fwd16 $S 2>/dev/null
  echo "== 3 stages"; R3M_BF16_NST3=1 python tools/conv_bench.py fwd16 $S 2>/dev/null
done
