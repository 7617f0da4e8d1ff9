#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
REPO=$(pwd)
A="--steps 4 --warmup 2 --no-cpu-baseline"
for i in 1 2; do for P in fp32 bf16; do
python bench.py --precision $P $A > gpurun_out/pv.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/pv.json')); print('plain ', '$P', d['value'], d['ms_per_step'])"
python bench.py --precision $P $A --no-kernel-timing > gpurun_out/pv.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/pv.json')); print('plain-notiming ', '$P', d['value'], d['ms_per_step'])"
rm -rf /tmp/kt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision $P $A > $REPO/gpurun_out/pv.json 2>/dev/null); python -c "import json; d=json.load(open('gpurun_out/pv.json')); print('traced', '$P', d['value'], d['ms_per_step'])"
done; done
rm -rf /tmp/kt
env | grep -i -E "^HSA|^HIP|^ROC|^AMD|^GPU" | head
