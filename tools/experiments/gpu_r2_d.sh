#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/experiments/debug_crop_fused.py 2>&1 | tail -20
timeout 900 python -m pytest tests/test_gpu_augment.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider --timeout 900 -x 2>&1 | tail -15
bash tools/gpu_r2_prof.sh r34c4 "--size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj" r50bf16 "--precision bf16"
