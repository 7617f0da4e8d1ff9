#!/bin/bash
# kernel-trace summaries (rocprofv3 --kernel-trace -> rocpd -> CSV) of chosen bench configurations; usage: gpu_r2_prof.sh tag "bench args" ...
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd)
while [ $# -ge 2 ]; do
  tag=$1; args=$2; shift 2
  rm -rf /tmp/kt
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/kt_${tag}_bench.json 2> $REPO/gpurun_out/kt_$tag.log)
  db=$(find /tmp/kt -name "*.db" | head -1)
  python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_$tag.csv | tail -1
  cut -c1-200 gpurun_out/kt_${tag}_bench.json
  head -25 gpurun_out/kernel_stats_$tag.csv | cut -c1-150
  rm -rf /tmp/kt
done
