#!/bin/bash
# A/B: K steps of 32 rows (3 blocks per CU) for the bf16 GEMM and weight-gradient kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
R3M_BF16_BK=32 R3M_WG16_BK=32 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "conv or wgrad or stem or encoder" > gpurun_out/bk32_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/bk32_tests.log
run() { env "$@" timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
run A=0
run R3M_BF16_BK=32
run R3M_WG16_BK=32
run R3M_BF16_BK=32 R3M_WG16_BK=32
run A=0
