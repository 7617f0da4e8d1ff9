#!/bin/bash
# round-5 experiment batch 2: (a) timing probes of the read-modify-write expanding dgrads; (b) early epilogue-operand prefetch
# (libr3m_hip_pfearly.so = -DR3M_PF_EARLY=1) against the shipped build, launch by launch and in the fp32 step; (c) launch CSVs of the
# fp32 headline and configs[2] for the two-roof report (tools/launch_report.py --roofs)
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_rmw_probe.sh > /dev/null
S="1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 1280,14,1024,256,1,1,0 1280,7,2048,512,1,1,0"
{
for rep in 1 2; do
  echo "== shipped rep $rep"; python tools/conv_bench.py dgradbnres $S 2>/dev/null | cut -c1-80
  echo "== pfearly rep $rep"; R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_pfearly.so python tools/conv_bench.py dgradbnres $S 2>/dev/null | cut -c1-80
done
echo "== dgradbn (EPI_BNRED only, y recomputed mask) shipped / pfearly"
python tools/conv_bench.py dgradbn $S 2>/dev/null | cut -c1-80
R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_pfearly.so python tools/conv_bench.py dgradbn $S 2>/dev/null | cut -c1-80
} 2>&1 | tee gpurun_out/r05_pfearly_ops.txt
B="timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3"
for rep in 1 2 3; do
  for v in base pfearly; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = pfearly ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_pfearly.so
    R3M_HIP_LIB=$LIB $B 2>/dev/null > gpurun_out/pf_c1_$v.json
    python - <<PY
import json
j = json.load(open("gpurun_out/pf_c1_$v.json"))
print("$v rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms class", j["roofline"]["frac"])
PY
  done
done 2>&1 | tee gpurun_out/r05_pfearly_step.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 3 --launch-csv gpurun_out/r05_launch_fp32.csv > gpurun_out/r05_launch_fp32.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 3 --precision bf16 --langweight 1 --launch-csv gpurun_out/r05_launch_bf16.csv > gpurun_out/r05_launch_bf16.json 2>/dev/null
python tools/launch_report.py gpurun_out/r05_launch_fp32.csv 10 --roofs fp32 > gpurun_out/r05_tworoof_fp32.txt
python tools/launch_report.py gpurun_out/r05_launch_bf16.csv 10 --roofs bf16 > gpurun_out/r05_tworoof_bf16.txt
