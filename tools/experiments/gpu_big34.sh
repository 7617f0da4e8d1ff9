#!/bin/bash
# 256x128 tile on the 3x3-dominated models (ResNet-34 / 18, bf16)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env $E timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --precision bf16 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', '$E', d['value'], d['ms_per_step'], [(k['kernel'][:18], round(k['ms_per_step'],2), round(k['tflops'])) for k in d['roofline']['kernels']])"; }
for E in "A=0" "R3M_BF16_BIG=32" "R3M_BF16_BIG=64" "A=1"; do
  run r34 --size 34 --clips-per-gpu 512
done
for E in "A=0" "R3M_BF16_BIG=32"; do
  run r18 --size 18 --clips-per-gpu 512
done
