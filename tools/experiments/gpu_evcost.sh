#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for P in fp32 bf16; do for F in "" "--no-kernel-timing"; do
timeout 900 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline $F > gpurun_out/ev.json 2> gpurun_out/ev.err
python -c "import json; d=json.load(open('gpurun_out/ev.json')); print('$P', '$F', d['value'], d['ms_per_step'])"
done; done
