#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for P in fp32 bf16 fp32 bf16; do
sleep 40
python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/cold.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/cold.json')); print('after 40 s idle, default prewarm: $P', d['value'], d['ms_per_step'], d['prewarm_steps'])"
done
