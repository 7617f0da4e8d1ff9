#!/bin/bash
# Same-box, same-library step A/B of the round-6 bf16 3x3 kernels (r3m_debug_set_conv3x3_bf16: 0 = per-tile halo kernels, 1 = conv_row16.hip)
# on configs[2] (ResNet-50 bf16 + language) and configs[4] (ResNet-34 bf16 rctraj, 512 clips). usage: gpu_row16_step_ab.sh [tag] [reps]
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-row16}; REPS=${2:-2}
run() {  # $1 = mode, rest = bench args
  local mode=$1; shift
  timeout 600 python - "$@" <<PY
import sys, runpy
from r3m_amd import _lib
_lib.lib().r3m_debug_set_conv3x3_bf16($mode)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary", "--steps", "15", "--prewarm-seconds", "3"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
}
for rep in $(seq $REPS); do
  for mode in 0 1; do
    run $mode --precision bf16 --langweight 1 2>/dev/null > gpurun_out/${TAG}_c2_m$mode.json
    run $mode --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj 2>/dev/null > gpurun_out/${TAG}_c4_m$mode.json
    python - <<PY
import json
for c in ("c2", "c4"):
    j = json.load(open(f"gpurun_out/${TAG}_{c}_m$mode.json"))
    print("mode $mode rep $rep", c, j["value"], "frames/s", j["ms_per_step"], "ms")
PY
  done
done 2>&1 | tee gpurun_out/${TAG}_step_ab.txt
