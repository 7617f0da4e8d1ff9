#!/bin/bash
# Evidence run (one box, back to back; kernel traces cover 20 steps = 2 warm-up + 18 timed, no pre-warm, no secondary workloads): the driver's default bench line (headline + secondary configs + CPU baseline), every
# BASELINE config as its own bench line, the launcher forms (torchrun 1 rank, self-spawn), kernel traces (fp32 / bf16 / configs[4])
# and PMC passes (fp32 / bf16). Outputs under gpurun_out/; tools/collect_profiles.py <tag> copies the summaries into profiles/<tag>_*.
# usage: gpu_evidence.sh [tests]     ("tests": also the whole GPU suite + smoke first)
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd)
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt; lscpu | grep "Model name" >> gpurun_out/gpu.txt
if [ "$1" = "tests" ]; then
  rm -f gpurun_out/parity.txt
  ( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -rA ) > gpurun_out/pytest_all.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_all.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_all.log | head
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
fi
B="timeout 1200 python bench.py"
$B --launch-csv gpurun_out/launches_fp32.csv > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err; echo "bench default (fp32 headline + secondary + cpu) rc=$?"
$B --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --launch-csv gpurun_out/launches_bf16.csv > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench bf16 rc=$?"
$B --precision bf16 --langweight 1 --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/cfg_c2_r50_lang_bf16.json 2>/dev/null
$B --langweight 1 --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/cfg_c3_r50_lang_fp32.json 2>/dev/null
$B --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/cfg_c4_r34_bs512_bf16_rctraj.json 2>/dev/null
$B --size 34 --clips-per-gpu 512 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/cfg_r34_bs512_fp32.json 2>/dev/null
$B --size 18 --clips-per-gpu 512 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/cfg_r18_bs512_bf16.json 2>/dev/null
$B --encoder-only-frames 256 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/cfg_enc256_fp32.json 2>/dev/null
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --langweight 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/cfg_c3_torchrun_1rank_rccl.json 2> gpurun_out/torchrun.err; echo "torchrun rc=$?"
$B --gpus 1 --force-launcher --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/cfg_c1_selfspawn_1rank_rccl.json 2> gpurun_out/selfspawn.err; echo "self-spawn rc=$?"
# board power and clocks while the fp32 / bf16 steps run (a separate run: the sampler shares the host with the launcher thread)
for t in "fp32:" "bf16:--precision bf16 --langweight 1"; do
  tag=${t%%:*}; args=${t#*:}
  ( $B $args --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/power_${tag}_bench.json 2>/dev/null ) &
  bp=$!
  : > gpurun_out/power_$tag.txt
  for i in $(seq 60); do
    kill -0 $bp 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';' >> gpurun_out/power_$tag.txt; echo >> gpurun_out/power_$tag.txt
    sleep 0.5
  done
  wait $bp
done
for t in "fp32:" "bf16:--precision bf16" "r34c4:--size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj"; do
  tag=${t%%:*}; args=${t#*:}
  rm -rf /tmp/kt
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt -o trace -- python $REPO/bench.py $args --steps 18 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > $REPO/gpurun_out/kt_${tag}_bench.json 2> $REPO/gpurun_out/kt_$tag.log)
  db=$(find /tmp/kt -name "*.db" | head -1)
  python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_$tag.csv | tail -1
  rm -rf /tmp/kt
done
bash tools/gpu_pmc.sh 256 fp32 "" > gpurun_out/pmc_fp32.log 2>&1
bash tools/gpu_pmc.sh 256 bf16 _bf16 > gpurun_out/pmc_bf16.log 2>&1
bash tools/gpu_pmc.sh 512 bf16 _bf16_r34 "--size 34 --doaug rctraj" > gpurun_out/pmc_bf16_r34.log 2>&1
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/bench_*.json') + glob.glob('gpurun_out/cfg_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d['roofline']
        print(f"{f.split('/')[-1]:48s} {d['value']:10.1f} fps {d['ms_per_step']:8.2f} ms  {r['bound']} frac {r['frac']:.3f} whole {r['whole_step_frac']:.3f}  {d['config'].get('collectives','')[:40]}")
    except Exception as e:
        print(f, 'unreadable', e)
PY
