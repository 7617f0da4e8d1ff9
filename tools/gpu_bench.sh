#!/bin/bash
# bench + rocprofv3 kernel-trace on the GPU box; outputs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
CL=${1:-256}
timeout 900 python bench.py --clips-per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_32.json 2> gpurun_out/bench_32.err
echo "bench32 rc=$?"; cat gpurun_out/bench_32.json; tail -3 gpurun_out/bench_32.err
timeout 1500 python bench.py --clips-per-gpu $CL --steps 5 --warmup 2 > gpurun_out/bench_$CL.json 2> gpurun_out/bench_$CL.err
echo "bench$CL rc=$?"; cat gpurun_out/bench_$CL.json; tail -3 gpurun_out/bench_$CL.err
cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$CL -o trace -- python $REPO/bench.py --clips-per-gpu $CL --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_$CL.log 2>&1
echo "rocprof rc=$?"
cd $REPO; find gpurun_out/prof_$CL -name "*stats*" | head; 
f=$(find gpurun_out/prof_$CL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# the raw trace is large: keep only stats
find gpurun_out/prof_$CL -name "*kernel_trace.csv" -size +20M -delete
