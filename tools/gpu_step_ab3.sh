#!/bin/bash
# Same-box step A/B of libr3m_hip_base.so (tools/build_ab.sh <ref>) against the working-tree build on the three single-GPU configs:
# headline (ResNet-50 fp32), configs[2] (ResNet-50 bf16 + language), configs[4] (ResNet-34 bf16 rctraj, 512 clips). usage: gpu_step_ab3.sh [tag] [reps]
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-ab3}; REPS=${2:-2}
for rep in $(seq $REPS); do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
    B="timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3"
    R3M_HIP_LIB=$LIB $B 2>/dev/null > gpurun_out/${TAG}_c1_$v.json
    R3M_HIP_LIB=$LIB $B --precision bf16 --langweight 1 2>/dev/null > gpurun_out/${TAG}_c2_$v.json
    R3M_HIP_LIB=$LIB $B --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj 2>/dev/null > gpurun_out/${TAG}_c4_$v.json
    python - <<PY
import json
for c in ("c1", "c2", "c4"):
    j = json.load(open(f"gpurun_out/${TAG}_{c}_$v.json"))
    print("$v rep $rep", c, j["value"], "frames/s", j["ms_per_step"], "ms")
PY
  done
done 2>&1 | tee gpurun_out/${TAG}_step_ab.txt
