mkdir -p gpurun_out; export TMPDIR=/tmp
T=wgw3
{
R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so R3M_WG_WIN=1 python tools/wgrad_win_check.py save /tmp/win.pt
R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so R3M_WG_WIN=0 python tools/wgrad_win_check.py save /tmp/row.pt
python tools/wgrad_win_check.py cmp /tmp/win.pt /tmp/row.pt
for v in base new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v != new ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_$v.so
  echo "== $v"
  R3M_HIP_LIB=$LIB python tools/conv_bench.py wgrad 1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1 2>/dev/null
done
} 2>&1 | tee gpurun_out/${T}_check.txt
for rep in 1 2; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v != new ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_$v.so
    R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3 --launch-csv gpurun_out/${T}_launches_$v.csv 2>gpurun_out/${T}_$v.err > gpurun_out/${T}_c1_$v.json
    python - <<PY
import json
j = json.load(open("gpurun_out/${T}_c1_$v.json"))
print("$v rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms", "class frac", j["roofline"]["frac"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_step_ab.txt
for v in base new; do python tools/launch_report.py gpurun_out/${T}_launches_$v.csv 15 > gpurun_out/${T}_launch_report_$v.txt 2>&1; done
