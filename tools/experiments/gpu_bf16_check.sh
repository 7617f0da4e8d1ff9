#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/parity.txt
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/bf16.log 2>&1; tail -4 gpurun_out/bf16.log
grep bf16 gpurun_out/parity.txt | cut -c1-330
for i in 1 2; do timeout 900 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
python -c "import json; d=json.load(open('gpurun_out/bench_bf16.json')); print('bf16', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'])"; done
