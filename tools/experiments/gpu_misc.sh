#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -k "stem_tail or golden or conv_bf16 or emulation" > gpurun_out/misc_ops.log 2>&1; tail -3 gpurun_out/misc_ops.log
L="1280,56,64,256,1,1,0 1280,56,64,64,1,1,0 1280,28,128,512,1,1,0"
for v in; do echo "R3M_BF16_SINGLE=$v"; R3M_BF16_SINGLE=$v timeout 300 python tools/conv_bench.py fwd16 $L 2>&1 | grep -v amdgpu.ids; done
rm -rf /tmp/kt
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/misc_bench.json 2> $REPO/gpurun_out/misc.log)
db=$(find /tmp/kt -name "*.db" | head -1)
python -c "import json; d=json.load(open('gpurun_out/misc_bench.json')); print('bf16', d['value'], d['ms_per_step'])"
python tools/rocpd_stats.py "$db" gpurun_out/misc_kernel_stats.csv | grep -E "pool|total kernel"
rm -rf /tmp/kt
