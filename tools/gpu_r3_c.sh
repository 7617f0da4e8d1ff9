#!/bin/bash
# round 3: buffer-addressed wgrad — op parity, then per-shape A/B against the HEAD build on the same box
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad" -p no:cacheprovider 2>&1 | tail -5
SH="1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1 1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,28,128,512,1,1,0 1280,28,512,128,1,1,0 1280,56,64,256,1,1,0 1280,56,256,64,1,1,0 1280,56,128,128,3,2,1 1280,56,256,512,1,2,0 1280,7,2048,512,1,1,0"
for rep in 1 2; do
echo "== base (HEAD)";   R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py wgrad $SH
echo "== new";           python tools/conv_bench.py wgrad $SH
echo "== new interleave"; R3M_WG_INTERLEAVE=1 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad $SH
done 2>&1 | tee gpurun_out/r3c_wgrad_ab.txt
