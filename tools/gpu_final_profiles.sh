#!/bin/bash
# round-end evidence run: all GPU tests, smoke, both bench lines (with launch CSVs), kernel traces and PMC passes for fp32 and bf16
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd)
rm -f gpurun_out/parity.txt
bash tools/gpu_run_tests.sh > gpurun_out/tests_tail.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 900 -p no:cacheprovider -n 3 > gpurun_out/bf16.log 2>&1; echo "bf16 rc=$?" | tee -a gpurun_out/summary.txt
tail -3 gpurun_out/ops.log; grep -aE "passed|failed|FAILED|crashed" gpurun_out/encoder.log | tail -4; tail -3 gpurun_out/bf16.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 1200 python bench.py --steps 5 --warmup 2 --launch-csv gpurun_out/launches_fp32.csv > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err; echo "bench fp32 rc=$?"
timeout 900 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/launches_bf16.csv > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench bf16 rc=$?"
for P in fp32 bf16; do
  rm -rf /tmp/kt
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/kt_${P}_bench.json 2> $REPO/gpurun_out/kt_$P.log)
  db=$(find /tmp/kt -name "*.db" | head -1)
  python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_$P.csv | tail -1
  rm -rf /tmp/kt
done
bash tools/gpu_pmc.sh 256 fp32 "" > gpurun_out/pmc_fp32.log 2>&1
bash tools/gpu_pmc.sh 256 bf16 _bf16 > gpurun_out/pmc_bf16.log 2>&1
python -c "
import json
for p in ('fp32','bf16'):
    d=json.load(open('gpurun_out/bench_%s.json'%p)); print(p, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d.get('cpu_baseline',{}).get('value'))"
ls gpurun_out | head -60
