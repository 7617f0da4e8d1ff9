mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --timeout 600 -p no:cacheprovider -k "kink_free or full_step" 2>&1 | grep -v Warning | tail -60 ) | tee gpurun_out/pw5_tests.txt
grep -n "kink-free\|full step" gpurun_out/parity.txt | tail -40
bash tools/gpu_pmc.sh 512 bf16 _bf16_r34 "--size 34 --doaug rctraj" > gpurun_out/pmc_bf16_r34.log 2>&1; tail -3 gpurun_out/pmc_bf16_r34.log
bash tools/gpu_pmc.sh 256 fp32 "" > gpurun_out/pmc_fp32.log 2>&1; tail -3 gpurun_out/pmc_fp32.log
