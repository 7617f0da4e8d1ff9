#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do
 for v in base new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 4 2>/dev/null > gpurun_out/tmp_$v.json
  python - <<PY
import json
j=json.load(open("gpurun_out/tmp_$v.json"))
print("$v", j["value"], j["ms_per_step"], [(k["kernel"][:18], round(k["ms_per_step"],2), round(k["tflops"],1)) for k in j["roofline"]["kernels"]])
PY
 done
done | tee gpurun_out/r3_step_ab_rows.txt
