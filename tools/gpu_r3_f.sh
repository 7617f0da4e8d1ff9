#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
S="2560,28,128,128,3,1,1 2560,14,256,256,3,1,1 2560,7,512,512,3,1,1 2560,28,128,256,3,2,1"
for rep in 1 2; do
echo "== new"; python tools/conv_bench.py wgrad16 $S
echo "== interleave"; R3M_WG_INTERLEAVE=1 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad16 $S
echo "== rows off"; R3M_WG16_ROWS=0 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad16 $S
echo "== blocks 2048"; R3M_WG16_BLOCKS=2048 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad16 $S
echo "== blocks 512"; R3M_WG16_BLOCKS=512 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py wgrad16 $S
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3f.txt
