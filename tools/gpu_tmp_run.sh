mkdir -p gpurun_out; export TMPDIR=/tmp
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
{
for v in 0 1; do
  echo "== R3M_PW_N128=$v"
  R3M_PW_N128=$v python tools/conv_bench.py fwd 1280,56,256,64,1,1,0 1280,56,64,64,1,1,0 2>/dev/null
  R3M_PW_N128=$v python tools/conv_bench.py dgradbn 1280,56,64,256,1,1,0 2>/dev/null
  R3M_PW_N128=$v python tools/conv_bench.py dgradbnres 1280,56,64,256,1,1,0 2>/dev/null
done
} 2>&1 | tee gpurun_out/n128_probe.txt
