#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -n 3 -k "bn" > gpurun_out/bn_ops.log 2>&1; tail -3 gpurun_out/bn_ops.log
S="4014080,64 4014080,256 1003520,512 250880,1024 62720,2048 250880,256"
for it in 1 2 4; do echo "R3M_BN_ITEMS=$it"; R3M_BN_ITEMS=$it timeout 300 python tools/bn_bench.py $S 2>&1 | grep -v amdgpu.ids; done
