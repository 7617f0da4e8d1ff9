S="2560,14,256,256,3,1,1 2560,28,128,128,3,1,1 2560,7,512,512,3,1,1 2560,56,64,64,3,1,1"
for i in 1 2; do
echo "== base (round-5 HEAD~: one zero row)"; PW16_MODE=0 R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_base.so python tools/conv_bench.py fwd16 $S 2>/dev/null | cut -c1-75
echo "== new (zero window)"; PW16_MODE=0 python tools/conv_bench.py fwd16 $S 2>/dev/null | cut -c1-75
done
