#!/bin/bash
# rocprofv3 kernel trace of a short bench run; only the per-kernel summary CSV is kept
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); CL=${1:-256}; TAG=${2:-prof}
rm -rf /tmp/kt
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --clips-per-gpu $CL --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/${TAG}_bench.json 2> $REPO/gpurun_out/${TAG}.log)
db=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" gpurun_out/${TAG}_kernel_stats.csv | head -40
rm -rf /tmp/kt
