#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stem" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for v in prev new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = prev ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 3 --launch-csv gpurun_out/launch_$v.csv 2>/dev/null > gpurun_out/tmp_$v.json
  echo "== $v $(python -c "import json; j=json.load(open('gpurun_out/tmp_$v.json')); print(j['value'], j['ms_per_step'])")"; python tools/launch_report.py gpurun_out/launch_$v.csv 10 | awk '$4==147'
done
done
