#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 3 --timeout 600 -p no:cacheprovider -k "conv or linear or stem" > gpurun_out/ops.log 2>&1; tail -4 gpurun_out/ops.log
for v in 1 0; do echo "R3M_WG_GLDS=$v"; R3M_WG_GLDS=$v python tools/conv_bench.py wgrad 1280,14,256,256,3,1,1 1280,14,256,1024,1,1,0 1280,28,128,128,3,1,1 1280,56,64,64,3,1,1 1280,56,64,256,1,1,0 1280,7,512,2048,1,1,0 2>&1 | grep -v amdgpu.ids; done
echo NARROW; for v in 2 0; do R3M_GG_GLDS=$v python tools/conv_bench.py fwd 1280,56,64,64,3,1,1 1280,56,256,64,1,1,0 2>&1 | grep -v amdgpu.ids; done
