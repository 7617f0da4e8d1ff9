#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --timeout 900 -x -k "bn_backward_partials" -n 4 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_bf16.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | tail -12
PROBES=$(pwd)/r3m_amd/lib/variants/libr3m_hip_probes.so
for k in 0 1 0 1; do
  echo "R3M_BNRED=$k fp32"; R3M_HIP_LIB=$PROBES R3M_BNRED=$k timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c80-260
done
for k in 0 1 0 1; do
  echo "R3M_BNRED=$k bf16"; R3M_HIP_LIB=$PROBES R3M_BNRED=$k timeout 600 python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c80-260
done
for k in 0 1; do
  echo "R3M_BNRED=$k r34 bf16"; R3M_HIP_LIB=$PROBES R3M_BNRED=$k timeout 600 python bench.py --size 34 --clips-per-gpu 512 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c80-260
done
