#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 900 -p no:cacheprovider -k "encoder or train" > gpurun_out/bf16.log 2>&1; tail -5 gpurun_out/bf16.log
cat gpurun_out/parity.txt 2>/dev/null
