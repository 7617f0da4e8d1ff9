#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad" -p no:cacheprovider 2>&1 | tail -3
S="1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1 1280,56,128,128,3,2,1 1280,28,256,256,3,2,1"
for rep in 1 2; do
echo "== rows (new)"; python tools/conv_bench.py wgrad $S
echo "== per-tap"; R3M_WG_ROWS=0 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad $S
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_wgrad_rows_ab.txt
