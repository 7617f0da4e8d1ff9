#!/bin/bash
# bf16 step: bench line (with launch CSV) + rocprofv3 kernel trace summary
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python bench.py --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/launches_bf16.csv > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_bf16.err
python -c "import json; d=json.load(open('gpurun_out/bench_bf16.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac']); [print(k) for k in d['roofline']['kernels']]"
rm -rf /tmp/kt
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof16_bench.json 2> $REPO/gpurun_out/prof16.log)
db=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" gpurun_out/prof16_kernel_stats.csv | head -45
rm -rf /tmp/kt
