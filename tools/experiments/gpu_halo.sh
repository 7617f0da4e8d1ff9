#!/bin/bash
# halo-tile 3x3 kernel: parity under the switch, isolated launches, whole step
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 > gpurun_out/halo_tests.log 2>&1; echo "default (halo) bf16 tests rc=$?"; tail -3 gpurun_out/halo_tests.log
R3M_BF16_HALO=3 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 -k "conv or encoder" > gpurun_out/halo_tests3.log 2>&1; echo "halo=3 tests rc=$?"; tail -3 gpurun_out/halo_tests3.log
SH="1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,56,64,64,3,1,1 1280,7,512,512,3,1,1"
python tools/conv_bench.py fwd16 $SH > /dev/null
for H in 0 1 3; do echo "== R3M_BF16_HALO=$H"; R3M_BF16_HALO=$H python tools/conv_bench.py fwd16 $SH; done
run() { tag=$1; shift; env $E timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$E', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
for E in R3M_BF16_HALO=0 R3M_BF16_HALO=1 R3M_BF16_HALO=3; do run r50; done
for E in R3M_BF16_HALO=0 R3M_BF16_HALO=1 R3M_BF16_HALO=3; do run r34 --size 34 --clips-per-gpu 512; done
