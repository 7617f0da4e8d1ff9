#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
bash tools/gpu_run_tests.sh > gpurun_out/tests_tail.txt 2>&1
tail -3 gpurun_out/ops.log; grep -aE "passed|failed|FAILED|crashed" gpurun_out/encoder.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1200 python bench.py --steps 3 --warmup 1 --launch-csv gpurun_out/launches_256.csv > gpurun_out/bench_256.json 2> gpurun_out/bench_256.err
echo "bench rc=$?"; python -c "import json; d=json.load(open('gpurun_out/bench_256.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'], d.get('cpu_baseline',{}).get('value'))"
