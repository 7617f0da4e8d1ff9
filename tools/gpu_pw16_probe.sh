#!/bin/bash
# Where does a K step's time go in the persistent bf16 kernel (csrc/conv_pw16.hip)? Probe build (tools/build_ab.sh none probes).
# R3M_GG_DEBUG bits: 1 no result stores, 2 statistics only (no slab / stores), 4 no A DMA, 8 no B DMA, 16 no fragment reads / MFMAs.
# WRONG results with any bit. 19 = DMA + barriers only, 12 = everything but the DMA, 13 = MFMAs + slab only.
mkdir -p gpurun_out; export TMPDIR=/tmp
S="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,7,512,2048,1,1,0 1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 1280,14,1024,256,1,1,0 1280,7,2048,512,1,1,0 2560,14,256,256,3,1,1 2560,28,128,128,3,1,1"
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so PW16_MODE=3
for d in 0 19 12 13; do
  echo "== R3M_GG_DEBUG=$d"
  R3M_GG_DEBUG=$d python tools/conv_bench.py fwd16 $S 2>/dev/null | cut -c1-75
done 2>&1 | tee gpurun_out/pw16_probe.txt
