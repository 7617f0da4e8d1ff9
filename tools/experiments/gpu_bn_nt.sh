#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
S="4014080,64 4014080,256 1003520,512 250880,1024 62720,2048"
echo "default policy"; timeout 300 python tools/bn_bench.py $S 2>&1 | grep -v amdgpu.ids
echo "non-temporal streams"; R3M_HIP_LIB=$(pwd)/r3m_amd/lib_alt/libr3m_hip.so timeout 300 python tools/bn_bench.py $S 2>&1 | grep -v amdgpu.ids
echo "default policy (again)"; timeout 300 python tools/bn_bench.py $S 2>&1 | grep -v amdgpu.ids
for lib in "" "$(pwd)/r3m_amd/lib_alt/libr3m_hip.so"; do for P in fp32 bf16; do
R3M_HIP_LIB=$lib timeout 900 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/nt.json 2> gpurun_out/nt.err
python -c "import json; d=json.load(open('gpurun_out/nt.json')); print('lib=[$lib]', '$P', d['value'], d['ms_per_step'])"
done; done
