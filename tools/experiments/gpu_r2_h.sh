#!/bin/bash
# traversal-order A/B of the BatchNorm passes (R3M_BN_REV compile-time variants), whole-step bench, same box
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$(pwd)/r3m_amd/lib/variants
for rep in 1 2; do
for v in "" _rev1 _rev2 _rev3 _rev7; do
  for cfg in "fp32:" "bf16:--precision bf16" "r34bf16:--size 34 --clips-per-gpu 512 --precision bf16"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    ms=$(R3M_HIP_LIB=$V/libr3m_hip_probes$v.so timeout 600 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "variant probes$v $tag ms_per_step $ms"
  done
done
done
