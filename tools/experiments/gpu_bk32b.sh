#!/bin/bash
# follow-up A/B after making 32-row K steps the default: per-shape launch tables, split target, side stream, fused stem tail (8-wide bf16)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 > gpurun_out/bk32b_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/bk32b_tests.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/l_$tag.csv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
run default A=0
run bk64 R3M_BF16_BK=64 R3M_WG16_BK=64
run wg1536 R3M_WG16_BLOCKS=1536
run wg2048 R3M_WG16_BLOCKS=2048
run side R3M_SIDE_STREAM=1
run default2 A=0
