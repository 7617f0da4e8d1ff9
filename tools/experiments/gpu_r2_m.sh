#!/bin/bash
export TMPDIR=/tmp
V=$(pwd)/r3m_amd/lib/variants
for rep in 1 2 3; do for v in _wg768 _pre; do
  ms=$(R3M_HIP_LIB=$V/libr3m_hip_probes$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['frac'], [round(k['ms_per_step'],1) for k in r['kernels']])")
  echo "variant probes$v fp32 $ms"
done; done
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider --timeout 600 -k "stem" 2>&1 | tail -2
