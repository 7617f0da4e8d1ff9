#!/bin/bash
# per-shape A/B of the fp32 weight-gradient kernels: libr3m_hip_base.so (tools/build_ab.sh <ref>) against the working tree
S="1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,28,128,512,1,1,0 1280,28,512,128,1,1,0 1280,56,64,256,1,1,0 1280,56,256,64,1,1,0 1280,56,64,64,1,1,0 1280,7,512,2048,1,1,0 1280,7,2048,512,1,1,0 1280,56,256,512,1,2,0 1280,28,512,1024,1,2,0 1280,14,256,256,3,1,1 1280,56,64,64,3,1,1"
for rep in 1 2 3; do
  echo "== base"; R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py wgrad $S 2>/dev/null
  echo "== new"; python tools/conv_bench.py wgrad $S 2>/dev/null
done
