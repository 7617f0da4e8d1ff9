#!/bin/bash
# round 3: the new tests without -x (every failure in one call)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_fullsize.py tests/test_gpu_lang.py tests/test_gpu_train.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/r3b_tests.txt
grep -v "^$" gpurun_out/r3b_tests.txt | tail -70
