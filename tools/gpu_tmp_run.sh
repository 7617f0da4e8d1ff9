mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_fwd_dgrad_wgrad or dgrad_epilogue" --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -6
done 2>&1 | tee gpurun_out/stress_ops.txt
for rep in 1 2; do
  timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -6
done 2>&1 | tee -a gpurun_out/stress_ops.txt
