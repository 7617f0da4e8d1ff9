#!/bin/bash
# all-taps 3x3 weight-gradient kernel: parity under the switch, isolated launches, whole step
mkdir -p gpurun_out; export TMPDIR=/tmp
R3M_WG16_HALO=1 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -n 3 -k "conv and k3s1" > gpurun_out/wghalo_tests.log 2>&1; echo "wg halo=1 conv tests rc=$?"; tail -4 gpurun_out/wghalo_tests.log | cut -c1-220
SH="1280,56,64,64,3,1,1 2560,56,64,64,3,1,1"
python tools/conv_bench.py wgrad16 $SH > /dev/null
for H in 0 1; do echo "== R3M_WG16_HALO=$H"; R3M_WG16_HALO=$H python tools/conv_bench.py wgrad16 $SH; done
run() { tag=$1; shift; env $E timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$E', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
for E in R3M_WG16_HALO=0 R3M_WG16_HALO=1; do run r50; done
for E in R3M_WG16_HALO=0 R3M_WG16_HALO=1; do run r34 --size 34 --clips-per-gpu 512; done
