#!/bin/bash
# Round 4: correctness + A/B of the persistent 1x1 kernel (conv_pw.hip) against libr3m_hip_base.so (tools/build_ab.sh <ref>).
# usage: gpu_pw_ab.sh [tag] [notests]
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-pw}
if [ "$2" != "notests" ]; then
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad or dgrad_epilogue or transpose_safe" --timeout 600 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
echo "ops tests:"; tail -3 gpurun_out/${TAG}_tests.log
( timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests2.log 2>&1
echo "encoder tests:"; tail -3 gpurun_out/${TAG}_tests2.log
fi
S50="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,7,512,2048,1,1,0 1280,7,2048,512,1,1,0"
D50="1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 1280,14,1024,256,1,1,0"
E50="1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,7,512,2048,1,1,0"
for rep in 1 2; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
    echo "== $v"
    R3M_HIP_LIB=$LIB python tools/conv_bench.py fwd $S50 2>/dev/null
    R3M_HIP_LIB=$LIB python tools/conv_bench.py dgradbnres $D50 2>/dev/null
    R3M_HIP_LIB=$LIB python tools/conv_bench.py dgradbn $E50 2>/dev/null
  done
done 2>&1 | tee gpurun_out/${TAG}_shapes.txt
python - <<'PY' 2>&1 | tee gpurun_out/${TAG:-pw}_membw.txt
import torch
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
ms = t(lambda: x.fill_(1.0)); print(f"fill 4 GiB: {ms:.3f} ms  write {4.295/ms:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy 4 GiB: {ms:.3f} ms  read+write {2*4.295/ms:.2f} TB/s")
ms = t(lambda: x.sum()); print(f"sum 4 GiB: {ms:.3f} ms  read {4.295/ms:.2f} TB/s")
PY
for rep in 1 2; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
    R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3 --launch-csv gpurun_out/${TAG}_launches_$v.csv 2>/dev/null > gpurun_out/${TAG}_c1_$v.json
    python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_c1_$v.json"))
print("$v rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms", "class frac", j["roofline"]["frac"])
PY
  done
done 2>&1 | tee gpurun_out/${TAG}_step_ab.txt
for v in base new; do python tools/launch_report.py gpurun_out/${TAG}_launches_$v.csv 15 > gpurun_out/${TAG}_launch_report_$v.txt 2>&1; done
