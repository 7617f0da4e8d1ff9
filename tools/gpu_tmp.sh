#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for v in gather window; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so; [ $v = gather ] && export R3M_GG_WIN=0 || export R3M_GG_WIN=1
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 4 --launch-csv gpurun_out/launch_$v.csv 2>/dev/null > gpurun_out/tmp_$v.json
  echo "== $v"; python tools/launch_report.py gpurun_out/launch_$v.csv 10 | awk '$5==9 || NR==1' 
done
