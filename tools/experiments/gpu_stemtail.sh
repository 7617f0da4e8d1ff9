#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/parity.txt
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_train.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -k "stem_tail or golden or step or emulation or track or maxpool" > gpurun_out/st_ops.log 2>&1; tail -4 gpurun_out/st_ops.log
bash tools/gpu_bench_both.sh
