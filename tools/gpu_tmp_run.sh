mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_final
for rep in 1 2 3; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v != new ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_$v.so
    R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3 --launch-csv gpurun_out/${T}_launches_$v.csv 2>gpurun_out/${T}_$v.err > gpurun_out/${T}_c1_$v.json
    python - <<PY
import json
j = json.load(open("gpurun_out/${T}_c1_$v.json"))
print("$v rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms", "class frac", j["roofline"]["frac"], "whole", j["roofline"]["whole_step_frac"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_step_ab.txt
for v in base new; do python tools/launch_report.py gpurun_out/${T}_launches_$v.csv 15 > gpurun_out/${T}_launch_report_$v.txt 2>&1; done
for v in base new; do
  LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v = base ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 2 --size 34 --clips-per-gpu 512 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('r34 fp32 $v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['whole_step_frac'])"
  R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 10 --prewarm-seconds 2 --size 18 --clips-per-gpu 512 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('r18 fp32 $v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['whole_step_frac'])"
done 2>&1 | tee -a gpurun_out/${T}_step_ab.txt
