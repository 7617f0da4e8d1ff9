#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/experiments/debug_repeat2.py fp32 15 18 2>&1 | grep -v amdgpu.ids
python tools/experiments/debug_repeat2.py fp32 16 18 2>&1 | grep -v amdgpu.ids
python tools/experiments/debug_repeat2.py bf16 16 18 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_augment.py -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | tail -8
