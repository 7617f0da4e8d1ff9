#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_augment.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
python tools/diag_secondary2.py base 2>&1 | grep -v amdgpu | grep "configs\[4\]" | tr "\n" " "; echo
for i in 1 2 3; do
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/tmp_def_$i.json 2>gpurun_out/tmp_def_$i.err
python - <<PY
import json
j=json.load(open("gpurun_out/tmp_def_$i.json"))
print($i, j["value"], j["ms_per_step"], {k:(v.get("value"), v.get("ms_per_step")) for k,v in j["secondary"].items()})
PY
done
