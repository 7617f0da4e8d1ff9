mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -m gpu -q -k "dgrad_epilogue or fused_and_standalone or full_step" --timeout 600 -p no:cacheprovider 2>&1 | grep -v Warning | tail -6 )
grep -n "post-step" gpurun_out/parity.txt | tail -12
P=$PWD/r3m_amd/lib/libr3m_hip_probes.so
for rep in 1 2; do
  for v in "bnred1:R3M_BNRED=1" "bnred2:R3M_BNRED=2"; do
    tag=${v%%:*}; envs=${v#*:}
    env $envs R3M_HIP_LIB=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3 --precision bf16 --langweight 1 2>/dev/null > gpurun_out/bn16_c2_$tag.json
    env $envs R3M_HIP_LIB=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 15 --prewarm-seconds 3 --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj 2>/dev/null > gpurun_out/bn16_c4_$tag.json
    python - <<PY
import json
for c in ("c2", "c4"):
    j = json.load(open(f"gpurun_out/bn16_{c}_$tag.json"))
    print("$tag rep $rep", c, j["value"], "frames/s", j["ms_per_step"], "ms")
PY
  done
done 2>&1 | tee gpurun_out/bn16_ab.txt
