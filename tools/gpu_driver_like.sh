#!/bin/bash
# what the driver does at round end: the GPU test tier in one serial pytest process, smoke(), the default bench line
export TMPDIR=/tmp; mkdir -p gpurun_out
(time timeout 2200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider) > gpurun_out/serial_gpu.log 2>&1; tail -4 gpurun_out/serial_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
(time python bench.py) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -4 gpurun_out/bench_default.err
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(json.dumps({k:v for k,v in d.items() if k not in ('roofline','config')})); print(d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'])"
(time python bench.py --precision bf16 --no-cpu-baseline) > gpurun_out/bench_default_bf16.json 2> gpurun_out/bench_default_bf16.err; python -c "import json; d=json.load(open('gpurun_out/bench_default_bf16.json')); print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'])"
