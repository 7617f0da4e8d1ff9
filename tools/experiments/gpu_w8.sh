#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
L="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,14,1024,256,1,1,0 1280,14,256,256,3,1,1 1280,56,64,64,3,1,1 1280,56,256,64,1,1,0 1280,28,128,128,3,1,1 1280,7,2048,512,1,1,0 1280,28,512,128,1,1,0"
for v in 0 -1 0 -1; do echo "R3M_BF16_RING=$v"; R3M_BF16_RING=$v timeout 300 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids; done
