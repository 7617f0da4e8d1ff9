#!/bin/bash
# A/B of env-switchable variants on the headline bench (3 steps each), interleaved twice
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
  for v in "R3M_SIDE_STREAM=1" "R3M_SIDE_STREAM=0"; do
    env $v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('$v'.ljust(24), d['value'], d['ms_per_step'], d['roofline']['frac'], [round(k['ms_per_step'],1) for k in d['roofline']['kernels']])"
  done
done
