#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -n 3 -k "conv" > gpurun_out/ring_ops.log 2>&1; tail -3 gpurun_out/ring_ops.log
L="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,14,1024,256,1,1,0 1280,14,256,256,3,1,1 1280,56,64,64,3,1,1 1280,56,256,64,1,1,0 1280,28,128,128,3,1,1 1280,7,512,512,3,1,1 1280,7,2048,512,1,1,0 1280,28,512,128,1,1,0"
for v in 0 3 1; do echo "R3M_BF16_RING=$v"; R3M_BF16_RING=$v timeout 300 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids; done
