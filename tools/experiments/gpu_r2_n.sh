#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider --timeout 900 -n 4 -k "bn or encoder or stem or fused" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fp32', d['value'], d['ms_per_step'], r['frac'], r['whole_step_frac'])"; done
