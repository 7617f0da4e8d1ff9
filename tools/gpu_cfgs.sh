#!/bin/bash
# the other BASELINE configurations on the current build, one bench line each -> gpurun_out/cfg_*.json
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/cfg_$tag.json 2> gpurun_out/cfg_$tag.err; echo "$tag rc=$?"
  python -c "import json; d=json.load(open('gpurun_out/cfg_$tag.json')); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], r['bound'], r['frac'], r.get('whole_step_frac'))"; }
run c2_r50_lang_fp32 --langweight 1.0
run c2_r50_lang_bf16 --langweight 1.0 --precision bf16
run r34_bs512_fp32 --size 34 --clips-per-gpu 512
run c4_r34_bs512_bf16 --size 34 --clips-per-gpu 512 --precision bf16
run c4_r34_bs512_bf16_rctraj --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj
run r18_bs512_bf16 --size 18 --clips-per-gpu 512 --precision bf16
run r18_bs512_fp32 --size 18 --clips-per-gpu 512
run enc256_fp32 --encoder-only-frames 256
run enc256_bf16 --encoder-only-frames 256 --precision bf16
