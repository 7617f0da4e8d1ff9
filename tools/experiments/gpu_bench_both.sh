#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for P in fp32 bf16; do
timeout 900 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/launches_$P.csv > gpurun_out/bench_$P.json 2> gpurun_out/bench_$P.err; echo "bench $P rc=$?"
python -c "import json; d=json.load(open('gpurun_out/bench_$P.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac']); [print(' ', k['kernel'][:28], round(k['ms_per_step'],2), round(k['tflops'],1)) for k in d['roofline']['kernels']]"
done
