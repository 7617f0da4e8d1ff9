#!/bin/bash
# Where does a short-K tile's time go? Timing probes of the persistent 1x1 kernel (probe build: tools/build_ab.sh none probes).
# R3M_GG_DEBUG bits: 1 no stores, 2 no statistics, 4 no epilogue at all, 8 no DMA (stale LDS). Results are WRONG with any bit set.
mkdir -p gpurun_out; export TMPDIR=/tmp
S="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,7,512,2048,1,1,0"
for d in 0 1 2 3 4 8 12; do
  echo "== R3M_GG_DEBUG=$d"
  R3M_GG_DEBUG=$d R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py fwd $S 2>/dev/null
done 2>&1 | tee gpurun_out/pw_probe.txt
