#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -n 3 -k "conv or stem or linear" > gpurun_out/ops32.log 2>&1; tail -2 gpurun_out/ops32.log
L="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,14,256,256,3,1,1 1280,56,64,64,1,1,0 1280,56,256,64,1,1,0 1280,14,1024,256,1,1,0"
timeout 300 python tools/conv_bench.py fwd $L 2>&1 | grep -v amdgpu.ids
