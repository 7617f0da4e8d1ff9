#!/bin/bash
# traversal-order A/B with the non-temporal load policy of the BatchNorm streams switched off (R3M_BN_NT=0 variants)
export TMPDIR=/tmp
V=$(pwd)/r3m_amd/lib/variants
for rep in 1 2; do
for v in "" _nt0 _nt0rev3 _nt0rev7; do
  for cfg in "fp32:" "bf16:--precision bf16"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    ms=$(R3M_HIP_LIB=$V/libr3m_hip_probes$v.so timeout 600 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "variant probes$v $tag ms_per_step $ms"
  done
done
done
