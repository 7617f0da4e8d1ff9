#!/bin/bash
export TMPDIR=/tmp
V=$(pwd)/r3m_amd/lib/variants
for rep in 1 2 3; do for v in "" _pre; do
  for cfg in "bf16:--precision bf16" "r34bf16:--size 34 --clips-per-gpu 512 --precision bf16"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    ms=$(R3M_HIP_LIB=$V/libr3m_hip_probes$v.so timeout 600 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "variant probes$v $tag ms_per_step $ms"
  done
done; done
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider --timeout 900 -n 4 2>&1 | tail -3
