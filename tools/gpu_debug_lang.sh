#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R3M_TRACE=1 timeout 600 python -m pytest "tests/test_gpu_lang.py::test_langrew_c_abi_vs_torch" -m gpu -q -x -s --timeout 300 -p no:cacheprovider > gpurun_out/lang_small.log 2>&1
echo "small rc=$?"; grep -vE "^\[r3m\]" gpurun_out/lang_small.log | tail -30; grep -E "^\[r3m\]" gpurun_out/lang_small.log | tail -4
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_lang.py -m gpu -q -s -n 2 --timeout 600 -p no:cacheprovider > gpurun_out/encoder.log 2>&1
grep -E "passed|failed|FAILED|grad-norm|crashed|vs fp64" gpurun_out/encoder.log | tail -40
