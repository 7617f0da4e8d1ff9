mkdir -p gpurun_out; export TMPDIR=/tmp
T=wgw2
{
python tools/experiments/wgw_debug.py 2>&1 | grep -E "case|kh" | tee gpurun_out/wgw2_debug.txt
R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so R3M_WG_WIN=1 python tools/wgrad_win_check.py save /tmp/win.pt
R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so R3M_WG_WIN=0 python tools/wgrad_win_check.py save /tmp/row.pt
python tools/wgrad_win_check.py cmp /tmp/win.pt /tmp/row.pt
echo "== microbench (probe build): window vs kernel rows"
for w in 1 0; do
  echo "R3M_WG_WIN=$w"
  R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so R3M_WG_WIN=$w python tools/conv_bench.py wgrad 1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1 2>/dev/null
done
} 2>&1 | tee gpurun_out/${T}_check.txt
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "wgrad or fuzz or conv_fwd" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -12 ) | tee gpurun_out/${T}_tests.txt
( timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_configs.py -m gpu -q --timeout 1200 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -8 ) | tee -a gpurun_out/${T}_tests.txt
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
for v in base new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 2 --size 34 --clips-per-gpu 512 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('r34 fp32 $v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done 2>&1 | tee -a gpurun_out/${T}_step_ab.txt
