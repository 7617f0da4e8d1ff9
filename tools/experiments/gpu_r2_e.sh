#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for p in fp32 bf16; do python tools/experiments/debug_repeat.py $p 15 2>&1 | grep -v amdgpu.ids; done
python tools/experiments/debug_repeat.py fp32 16 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_augment.py tests/test_gpu_configs.py tests/test_gpu_prefetch.py -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | tail -15
