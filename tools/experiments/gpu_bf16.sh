#!/bin/bash
# bf16 path: per-op + engine tests, then single-launch timings of representative ResNet-50 layers (bs 1280 frames)
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/bf16.log 2>&1; tail -25 gpurun_out/bf16.log
cat gpurun_out/parity.txt 2>/dev/null
L="1280,14,256,256,3,1,1 1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,28,128,512,1,1,0 1280,28,128,128,3,1,1 1280,56,64,64,3,1,1 1280,56,64,256,1,1,0 1280,56,256,64,1,1,0 1280,7,512,512,3,1,1 1280,7,512,2048,1,1,0"
timeout 600 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/conv_bench.py wgrad16 $L 2>&1 | grep -v amdgpu.ids
