#!/bin/bash
# GPU idle gaps of a short traced bench run: usage gpu_gaps.sh <tag> <bench args...>; writes gpurun_out/gaps_<tag>.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); TAG=$1; shift
rm -rf /tmp/kt_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$TAG -o trace -- python $REPO/bench.py "$@" --steps 6 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1)
DB=$(find /tmp/kt_$TAG -name "*.db" | head -1)
python tools/rocpd_gaps.py "$DB" 700 > gpurun_out/gaps_$TAG.txt 2>&1
head -36 gpurun_out/gaps_$TAG.txt | cut -c1-160
