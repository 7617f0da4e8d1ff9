#!/bin/bash
# round 3, first GPU call: the new tests (multi-GPU-armed DDP tests at world 1, headline backward chunks, mid-size oracle), then the default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
timeout 2400 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_fullsize.py tests/test_gpu_lang.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r3a_tests.txt
tail -30 gpurun_out/r3a_tests.txt
timeout 900 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
echo "bench rc=$?"; cut -c1-3000 gpurun_out/r3a_bench.json; tail -5 gpurun_out/r3a_bench.err
