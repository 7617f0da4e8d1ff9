#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_encoder.py -m gpu -q --timeout 600 -p no:cacheprovider -k "conv or stem or golden or bn or step" > gpurun_out/red_ops.log 2>&1; tail -2 gpurun_out/red_ops.log
for P in bf16 fp32; do
rm -rf /tmp/kt
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/red_${P}_bench.json 2> $REPO/gpurun_out/red_$P.log)
db=$(find /tmp/kt -name "*.db" | head -1)
python -c "import json; d=json.load(open('gpurun_out/red_${P}_bench.json')); print('$P', d['value'], d['ms_per_step'])"
python tools/rocpd_stats.py "$db" gpurun_out/red_kernel_stats_$P.csv | grep -E "wgrad_reduce|bn_stats_reduce|bn_finalize|bn_bwd_finalize|transpose_w|total kernel"
rm -rf /tmp/kt
done
