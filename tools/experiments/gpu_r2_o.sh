#!/bin/bash
export TMPDIR=/tmp
PROBES=$(pwd)/r3m_amd/lib/variants/libr3m_hip_probes.so
for rep in 1 2; do for k in 1 2; do
  R3M_HIP_LIB=$PROBES R3M_GG_K16=$k timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('R3M_GG_K16=$k', d['ms_per_step'], r['frac'], [round(k['ms_per_step'],1) for k in r['kernels']])"
done; done
R3M_HIP_LIB=$PROBES R3M_GG_K16=2 timeout 600 python tools/conv_bench.py fwd 1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,14,1024,256,1,1,0 1280,7,512,512,3,1,1
R3M_HIP_LIB=$PROBES R3M_GG_K16=1 timeout 600 python tools/conv_bench.py fwd 1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,14,1024,256,1,1,0 1280,7,512,512,3,1,1
