#!/bin/bash
# A/B: tiles per block (store acks overlap the next tile), 256x128 block tile for the main-loop-bound launches
mkdir -p gpurun_out; export TMPDIR=/tmp
for E in "R3M_BF16_TPB=3" "R3M_BF16_BIG=-1" "R3M_BF16_BIG=-1 R3M_BF16_TPB=2"; do
  env $E timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -n 3 -k "conv or encoder" > gpurun_out/tpb_tests.log 2>&1; echo "$E tests rc=$?"; tail -2 gpurun_out/tpb_tests.log
done
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --launch-csv gpurun_out/l_$tag.csv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:22], round(k['ms_per_step'],2)) for k in d['roofline']['kernels']])"; }
run base A=0
run tpb2 R3M_BF16_TPB=2
run tpb4 R3M_BF16_TPB=4
run tpb8 R3M_BF16_TPB=8
run big32 R3M_BF16_BIG=32
run big64 R3M_BF16_BIG=64
run base2 A=0
