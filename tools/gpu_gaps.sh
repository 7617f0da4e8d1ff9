#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
REPO=$(pwd)
for P in bf16 fp32; do
rm -rf /tmp/kt
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/gaps_${P}_bench.json 2> $REPO/gpurun_out/gaps_$P.log)
db=$(find /tmp/kt -name "*.db" | head -1)
python -c "import json; d=json.load(open('gpurun_out/gaps_${P}_bench.json')); print('$P', d['value'], d['ms_per_step'])"
python tools/rocpd_gaps.py "$db" > gpurun_out/gaps_$P.txt; head -48 gpurun_out/gaps_$P.txt
rm -rf /tmp/kt
done
