S34="2560,28,128,128,3,1,1 2560,14,256,256,3,1,1 2560,7,512,512,3,1,1 2560,56,64,128,3,2,1 2560,28,128,256,3,2,1 2560,14,256,512,3,2,1"
S50="1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 1280,56,128,128,3,2,1"
for rep in 1 2; do
echo "== base"; R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py wgrad16 $S34 $S50 2>/dev/null
echo "== new"; python tools/conv_bench.py wgrad16 $S34 $S50 2>/dev/null
done
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider -k "wgrad or geometry or conv" 2>&1 | tail -3
