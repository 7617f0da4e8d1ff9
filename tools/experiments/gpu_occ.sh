#!/bin/bash
# A/B of library variants (register budget -> blocks per CU) for the bf16 GEMM; variants from tools/experiments/build_variants.sh
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$(pwd)/r3m_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 > gpurun_out/occ_tests.log 2>&1; echo "default tests rc=$?"; tail -2 gpurun_out/occ_tests.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/l_$tag.csv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
run prev R3M_HIP_LIB=$V/libr3m_hip_prev.so
run w0n0 R3M_HIP_LIB=$V/libr3m_hip_w0n0.so
run w3n3 R3M_HIP_LIB=$V/libr3m_hip_w3n3.so
run w4n3 A=0
run w4n4 R3M_HIP_LIB=$V/libr3m_hip_w4n4.so
run w4n3bk32 R3M_BF16_BK=32
run prev2 R3M_HIP_LIB=$V/libr3m_hip_prev.so
