#!/bin/bash
# secondary BASELINE configs, one bench line each (no CPU baseline)
export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/cfg_$tag.json 2> gpurun_out/cfg_$tag.err; echo "$tag rc=$?"; python -c "import json; d=json.load(open('gpurun_out/cfg_$tag.json')); print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'], d['config']['workload'][:80])"; }
run c2_r50_lang_bf16 --precision bf16 --langweight 1.0 --steps 4 --warmup 2
run c2_r50_lang_fp32 --langweight 1.0 --steps 3 --warmup 1
run c4_r34_bs512_bf16 --precision bf16 --size 34 --clips-per-gpu 512 --steps 4 --warmup 2
run r34_bs512_fp32 --size 34 --clips-per-gpu 512 --steps 3 --warmup 1
run r18_bs512_bf16 --precision bf16 --size 18 --clips-per-gpu 512 --steps 4 --warmup 2
