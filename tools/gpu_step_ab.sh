#!/bin/bash
# round 3: full-step A/B (HEAD build vs working tree) + the parity tiers that exercise wgrad
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
    R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 4 > gpurun_out/r3d_${v}_$rep.json 2>/dev/null
    python - <<PY
import json
j=json.load(open("gpurun_out/r3d_${v}_$rep.json"))
print("$v $rep", j["value"], j["ms_per_step"], [(k["kernel"][:18], round(k["ms_per_step"],2), round(k["tflops"],1)) for k in j["roofline"]["kernels"]])
PY
  done
done 2>&1 | tee gpurun_out/r3d_step_ab.txt
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider -k "not bf16" 2>&1 | tail -8 | tee gpurun_out/r3d_tests.txt
