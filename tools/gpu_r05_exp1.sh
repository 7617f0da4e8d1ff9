#!/bin/bash
# round-5 experiment batch 1 (probe build = environment switches live):
#   (a) in-step per-shape report of the persistent bf16 kernel (configs[2] / configs[4])
#   (b) fp32 headline with the weight gradients on a side stream: R3M_SIDE_STREAM = 0 (off) / 1 (after dgrad) / 2 (beside dgrad)
#   (c) the same for configs[2]; R3M_PW16 = 0 / 1 for configs[2]
#   (d) golden tests under R3M_SIDE_STREAM=2
mkdir -p gpurun_out; export TMPDIR=/tmp
PROBES=$PWD/r3m_amd/lib/libr3m_hip_probes.so
{
echo "== (a) pw16 in-step report, ResNet-50 256 clips"
timeout 300 python tools/pw16_check.py report 50 256 2>/dev/null
echo "== (a) pw16 in-step report, ResNet-34 512 clips"
timeout 300 python tools/pw16_check.py report 34 512 2>/dev/null
} > gpurun_out/r05_pw16_instep_v5.txt 2>&1
B="timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3"
for rep in 1 2; do
  for m in 0 2 1; do
    R3M_HIP_LIB=$PROBES R3M_SIDE_STREAM=$m $B 2>/dev/null > gpurun_out/side_c1_$m.json
    python - <<PY
import json
j = json.load(open("gpurun_out/side_c1_$m.json"))
print("side=$m rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms class", j["roofline"]["frac"])
PY
  done
done 2>&1 | tee gpurun_out/r05_side_ab.txt
for rep in 1 2; do
  for v in "0 1" "2 1" "0 0" "2 0"; do
    set -- $v
    R3M_HIP_LIB=$PROBES R3M_SIDE_STREAM=$1 R3M_PW16=$2 $B --precision bf16 --langweight 1 2>/dev/null > gpurun_out/side_c2.json
    python - <<PY
import json
j = json.load(open("gpurun_out/side_c2.json"))
print("side=$1 pw16=$2 rep $rep c2", j["value"], "frames/s", j["ms_per_step"], "ms")
PY
  done
done 2>&1 | tee -a gpurun_out/r05_side_ab.txt
R3M_HIP_LIB=$PROBES R3M_SIDE_STREAM=2 timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05_side2_tests.txt
