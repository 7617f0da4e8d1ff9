#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 3 --timeout 600 -p no:cacheprovider -k "conv or linear or stem" > gpurun_out/ops.log 2>&1; tail -3 gpurun_out/ops.log
for v in 0 8 0 8; do echo "R3M_GG_DEBUG=$v (0 = interleaved DMA, 8 = clustered)"; R3M_GG_DEBUG=$v python tools/conv_bench.py fwd 1504,14,256,256,3,1,1 1280,14,256,256,3,1,1 1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,64,3,1,1 2>&1 | grep -v amdgpu.ids; done
