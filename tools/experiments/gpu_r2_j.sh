#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider --timeout 900 -k "fused_and_standalone" 2>&1 | tail -4; grep "fused vs" gpurun_out/parity.txt
PROBES=$(pwd)/r3m_amd/lib/variants/libr3m_hip_probes.so
for rep in 1 2; do for k in 0 2; do
  for cfg in "bf16:--precision bf16" "r34bf16:--size 34 --clips-per-gpu 512 --precision bf16"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    ms=$(R3M_HIP_LIB=$PROBES R3M_BNRED=$k timeout 600 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "R3M_BNRED=$k $tag ms_per_step $ms"
  done
done; done
