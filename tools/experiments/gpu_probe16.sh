#!/bin/bash
export TMPDIR=/tmp
L="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,14,1024,256,1,1,0 1280,14,256,256,3,1,1"
for v in 0 1 2 3; do echo "R3M_GG_DEBUG=$v (1: no epilogue, 2: no DMA after tile 0, 3: both)"; R3M_GG_DEBUG=$v timeout 300 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids; done
