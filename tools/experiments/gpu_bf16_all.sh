#!/bin/bash
# bf16 path: all bf16 tests, then the bf16 bench line + per-kernel trace summary
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
timeout 1800 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 900 -p no:cacheprovider -n 3 > gpurun_out/bf16.log 2>&1; tail -6 gpurun_out/bf16.log
cat gpurun_out/parity.txt 2>/dev/null
bash tools/gpu_prof16.sh
