mkdir -p gpurun_out; export TMPDIR=/tmp
for spin in 0 500 1000 2000; do ./tools/micro/storepat $spin; done 2>&1 | tee gpurun_out/storepat.txt
( timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 1200 -p no:cacheprovider -k "kink_free or full_step or bench_size" -s 2>&1 | grep -v Warning | tail -40 ) | tee gpurun_out/newparity.txt
