#!/bin/bash
# Where does a tile's time go in the persistent bf16 kernel (csrc/conv_pw16.hip)? Probe build (tools/build_ab.sh none probes).
# R3M_GG_DEBUG bits: 1 no result stores, 2 no hand-over (dump), 4 no A DMA, 8 no B DMA, 16 no fragment reads / MFMAs. WRONG results with any bit.
# R3M_PW16_ORDER: 0 tile order, 1 row-panel order.
mkdir -p gpurun_out; export TMPDIR=/tmp
S="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,7,512,2048,1,1,0 1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 1280,14,1024,256,1,1,0 1280,7,2048,512,1,1,0"
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
for o in 1; do
for d in 0 1 2 12 16 31; do
  echo "== R3M_PW16_ORDER=$o R3M_GG_DEBUG=$d"
  R3M_PW16_ORDER=$o R3M_GG_DEBUG=$d python tools/conv_bench.py fwd16 $S 2>/dev/null | cut -c1-75
done
done 2>&1 | tee gpurun_out/pw16_probe.txt
