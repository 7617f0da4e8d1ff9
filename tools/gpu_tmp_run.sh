mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_final_hot
# the driver's measurement conditions: 10 s pre-warm, 25 timed steps (bench.py defaults), round-3 build vs final build, interleaved
for rep in 1 2; do
  for v in base new; do
    LIB=$PWD/r3m_amd/lib/libr3m_hip.so; [ $v != new ] && LIB=$PWD/r3m_amd/lib/libr3m_hip_$v.so
    R3M_HIP_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>gpurun_out/${T}_$v.err > gpurun_out/${T}_c1_$v.json
    python - <<PY
import json
j = json.load(open("gpurun_out/${T}_c1_$v.json"))
print("$v rep $rep c1", j["value"], "frames/s", j["ms_per_step"], "ms", "class frac", j["roofline"]["frac"], "whole", j["roofline"]["whole_step_frac"], "steps", j["steps"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_step_ab.txt
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -k "H5_64to64 or H33_64to128 or H4_128to128" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6 ) | tee gpurun_out/${T}_newcases.txt
