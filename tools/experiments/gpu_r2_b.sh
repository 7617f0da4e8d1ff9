#!/bin/bash
# round-2 run B: new tests, K16 A/B per shape (probe build), fp32 bench, ResNet-34 bf16 rctraj fused vs unfused
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x ) > gpurun_out/pytest_b.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_b.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_b.log | head
PROBES=$(pwd)/r3m_amd/lib/variants/libr3m_hip_probes.so
SHAPES="1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,14,256,1024,1,1,0 1280,56,256,512,1,2,0 1280,28,512,1024,1,2,0 1280,7,512,2048,1,1,0"
for k in 0 1; do echo "R3M_GG_K16=$k"; R3M_HIP_LIB=$PROBES R3M_GG_K16=$k timeout 300 python tools/conv_bench.py fwd $SHAPES; done > gpurun_out/k16_ab.txt 2>&1
cat gpurun_out/k16_ab.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err
echo "bench fp32 rc=$?"; cut -c1-400 gpurun_out/bench_fp32.json
for u in "" "--unfused-crop"; do
  timeout 900 python bench.py --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj $u --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c4$u.json 2> gpurun_out/bench_c4$u.err
  echo "bench c4 $u rc=$?"; cut -c1-330 gpurun_out/bench_c4$u.json
done
timeout 900 python bench.py --size 34 --clips-per-gpu 512 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r34_nocrop.json 2>/dev/null; cut -c1-330 gpurun_out/bench_r34_nocrop.json
