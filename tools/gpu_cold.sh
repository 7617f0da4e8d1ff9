#!/bin/bash
# does a "cold" start (GPU idle for a while) measure slower than a warm one? bench with different warm-up lengths after idle periods
export TMPDIR=/tmp; mkdir -p gpurun_out
for W in 2 2 12 2 12; do
sleep 40
python bench.py --precision fp32 --steps 5 --warmup $W --no-cpu-baseline > gpurun_out/cold.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/cold.json')); print('after 40 s idle, warmup $W: fp32', d['value'], d['ms_per_step'])"
done
for W in 2 12; do
sleep 40
python bench.py --precision bf16 --steps 6 --warmup $W --no-cpu-baseline > gpurun_out/cold.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/cold.json')); print('after 40 s idle, warmup $W: bf16', d['value'], d['ms_per_step'])"
done
rocm-smi --showclocks 2>/dev/null | head -20
