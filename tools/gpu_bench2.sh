#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_run_tests.sh > gpurun_out/tests_tail.txt 2>&1
tail -3 gpurun_out/ops.log; tail -6 gpurun_out/encoder.log
timeout 1200 python bench.py --steps 3 --warmup 1 --launch-csv gpurun_out/launches_256.csv > gpurun_out/bench_256.json 2> gpurun_out/bench_256.err
echo "bench rc=$?"; cat gpurun_out/bench_256.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d.get('cpu_baseline'))"
