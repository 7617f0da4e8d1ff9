#!/bin/bash
# A/B: per-shape K tile rule (default) vs three-stage ring for the 32-wide K tiles
mkdir -p gpurun_out; export TMPDIR=/tmp
R3M_BF16_NST=7 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 -k "conv or encoder" > gpurun_out/nst3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/nst3_tests.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/l_$tag.csv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
run rule A=0
run bk32 R3M_BF16_BK=32
run nst3 R3M_BF16_NST=3
run nst7 R3M_BF16_NST=7
run nst3bk32 R3M_BF16_NST=3 R3M_BF16_BK=32
run rule2 A=0
