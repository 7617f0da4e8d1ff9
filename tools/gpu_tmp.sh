#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
S="2560,28,128,128,3,1,1 2560,14,256,256,3,1,1 2560,7,512,512,3,1,1 2560,56,64,64,3,1,1 1280,14,256,256,3,1,1 1280,56,64,64,3,1,1"
cp r3m_amd/lib/libr3m_hip_probes.so /tmp/prev.so 2>/dev/null
for rep in 1 2; do
echo "== new"; python tools/conv_bench.py fwd16 $S
echo "== prev (probes build of the previous commit)"; R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so python tools/conv_bench.py fwd16 $S
done 2>&1 | grep -v amdgpu.ids
for v in prev new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = prev ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/tmp_c4_$v.json
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --precision bf16 --langweight 1 --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/tmp_c2_$v.json
  python - <<PY
import json
for c in ("c4","c2"):
    j=json.load(open(f"gpurun_out/tmp_{c}_$v.json"))
    print("$v", c, j["value"], j["ms_per_step"], [(k["kernel"][:18], round(k["ms_per_step"],2), round(k["tflops"],1)) for k in j["roofline"]["kernels"]])
PY
done
