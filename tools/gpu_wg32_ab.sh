#!/bin/bash
# per-shape A/B of the fp32 weight-gradient kernels: libr3m_hip_base.so (tools/build_ab.sh <ref>) against the working tree
S="1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1 1280,56,128,128,3,2,1 1280,28,256,256,3,2,1 1280,14,512,512,3,2,1 1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0"
for rep in 1 2 3; do
  echo "== base"; R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py wgrad $S 2>/dev/null
  echo "== new"; python tools/conv_bench.py wgrad $S 2>/dev/null
done
