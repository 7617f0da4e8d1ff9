#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
ALT=$(pwd)/r3m_amd/lib_alt/libr3m_hip.so
L="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,14,256,256,3,1,1 1280,56,256,64,1,1,0"
for lib in "" "$ALT"; do echo "lib=[$lib]"; R3M_HIP_LIB=$lib timeout 300 python tools/conv_bench.py fwd $L 2>&1 | grep -v amdgpu.ids; R3M_HIP_LIB=$lib timeout 300 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids; done
for lib in "" "$ALT" "" "$ALT"; do for P in fp32 bf16; do
R3M_HIP_LIB=$lib timeout 900 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/nt.json 2> gpurun_out/nt.err
python -c "import json; d=json.load(open('gpurun_out/nt.json')); print('lib=[$lib]', '$P', d['value'], d['ms_per_step'])"
done; done
