#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -k "matches_reference_golden" -s -p no:cacheprovider 2>&1 | grep -v amdgpu | grep "r34 grad\|passed\|failed\|Error" | tail -14
for v in default bnred; do
  [ $v = bnred ] && export R3M_BNRED=2 || unset R3M_BNRED
  R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so timeout 600 python bench.py --no-cpu-baseline --precision bf16 --langweight 1 --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/tmp_c2_$v.json
  R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so timeout 600 python bench.py --no-cpu-baseline --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/tmp_c4_$v.json
  python - <<PY
import json
for c in ("c2","c4"):
    j=json.load(open(f"gpurun_out/tmp_{c}_$v.json"))
    print("$v", c, j["value"], j["ms_per_step"], [(k["kernel"][:18], round(k["ms_per_step"],2)) for k in j["roofline"]["kernels"]])
PY
done
