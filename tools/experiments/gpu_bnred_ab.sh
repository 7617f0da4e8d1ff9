#!/bin/bash
# EPI_BNRED on/off, fp32 ResNet-50 headline step, same box, alternating (probe build)
export TMPDIR=/tmp
PROBES=$(pwd)/r3m_amd/lib/variants/libr3m_hip_probes.so
for rep in 1 2 3; do for k in 0 1; do
  R3M_HIP_LIB=$PROBES R3M_BNRED=$k timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('R3M_BNRED=$k', d['value'], 'fps', d['ms_per_step'], 'ms; dominant class', r['achieved'], 'TF/s frac', r['frac'], 'whole', r['whole_step_frac'], [ (k['kernel'][:18], round(k['ms_per_step'],1)) for k in r['kernels']])"
done; done
