#!/bin/bash
# round 3: buffer-addressed bf16 wgrad — op parity, per-shape A/B vs the HEAD build (ResNet-34 at 2560 frames, ResNet-50 at 1280), configs[4]/[2] steps
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "not headline" -p no:cacheprovider 2>&1 | tail -4
S34="2560,28,128,128,3,1,1 2560,14,256,256,3,1,1 2560,7,512,512,3,1,1 2560,56,64,128,3,2,1 2560,28,128,256,3,2,1 2560,14,256,512,3,2,1 2560,56,64,128,1,2,0"
S50="1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,28,128,512,1,1,0 1280,28,512,128,1,1,0 1280,14,256,256,3,1,1 1280,56,256,512,1,2,0 1280,7,2048,512,1,1,0"
for rep in 1 2; do
echo "== base (HEAD)";   R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py wgrad16 $S34 $S50
echo "== new";           python tools/conv_bench.py wgrad16 $S34 $S50
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e_wgrad16_ab.txt
for v in base new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/r3e_c4_$v.json
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --precision bf16 --langweight 1 --steps 15 --prewarm-seconds 3 2>/dev/null > gpurun_out/r3e_c2_$v.json
  python - <<PY
import json
for c in ("c4","c2"):
    j=json.load(open(f"gpurun_out/r3e_{c}_$v.json"))
    print("$v", c, j["value"], j["ms_per_step"], [(k["kernel"][:18], round(k["ms_per_step"],2), round(k["tflops"],1)) for k in j["roofline"]["kernels"]])
PY
done 2>&1 | tee gpurun_out/r3e_step_ab.txt
