#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R3M_BF16_HALO=4 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 -k "conv" > gpurun_out/halo_tests4.log 2>&1; echo "halo=4 conv tests rc=$?"; tail -3 gpurun_out/halo_tests4.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "bf16" > gpurun_out/halo_full.log 2>&1; echo "fullsize bf16 rc=$?"; tail -3 gpurun_out/halo_full.log
SH="1280,7,512,512,3,1,1 1280,14,256,256,3,1,1"
python tools/conv_bench.py fwd16 $SH > /dev/null
for H in 1 5; do echo "== R3M_BF16_HALO=$H"; R3M_BF16_HALO=$H python tools/conv_bench.py fwd16 $SH; done
run() { tag=$1; shift; env $E timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$E', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
for E in R3M_BF16_HALO=1 R3M_BF16_HALO=5; do run r50; done
