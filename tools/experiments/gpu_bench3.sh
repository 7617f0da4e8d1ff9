#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/b3_$tag.json 2> gpurun_out/b3_$tag.err; echo "$tag rc=$?"; tail -2 gpurun_out/b3_$tag.err | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/b3_$tag.json')); r=d['roofline']
print(d['metric']); print(d['value'], d['ms_per_step'], d['dtype'], r['bound'], r['achieved'], r['unit'], r['frac'], r['whole_step_frac'], r.get('mfma'))
[print('  ', k['kernel'][:30], round(k['ms_per_step'],2), round(k['tflops'],1), round(k['algorithmic_GBps'],1)) for k in r['kernels']]"; }
run fp32 --steps 5 --warmup 2
run bf16 --precision bf16 --steps 6 --warmup 2
run enc256_fp32 --encoder-only-frames 256 --steps 10 --warmup 3
run enc256_bf16 --encoder-only-frames 256 --precision bf16 --steps 10 --warmup 3
run c4 --precision bf16 --size 34 --clips-per-gpu 512 --doaug rctraj --steps 4 --warmup 2
