#!/bin/bash
# Runs the GPU test tiers on the GPU box, one log per tier under gpurun_out/ (xdist isolates a crashing kernel to its test).
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt; lscpu | grep "Model name" >> gpurun_out/gpu.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 3 --timeout 600 -p no:cacheprovider > gpurun_out/ops.log 2>&1
echo "ops rc=$?" | tee -a gpurun_out/summary.txt
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_lang.py tests/test_gpu_augment.py tests/test_gpu_train.py tests/test_gpu_ddp.py -m gpu -q -s --timeout 900 -p no:cacheprovider > gpurun_out/encoder.log 2>&1
echo "encoder+lang rc=$?" | tee -a gpurun_out/summary.txt
tail -5 gpurun_out/ops.log; tail -30 gpurun_out/encoder.log
