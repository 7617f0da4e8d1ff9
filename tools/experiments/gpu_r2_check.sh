#!/bin/bash
# round-2 check: the whole GPU suite (one log), then the default bench line. Outputs under gpurun_out/.
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -rA ) > gpurun_out/pytest_all.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_all.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_all.log | head -20
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
