mkdir -p gpurun_out; export TMPDIR=/tmp
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
{
for d in 0 1 4 8 12; do
  echo "== narrow 3x3 (pw gather form, 256x64 tile) R3M_GG_DEBUG=$d"
  R3M_GG_DEBUG=$d python tools/conv_bench.py fwd 1280,56,64,64,3,1,1 1280,56,64,256,1,1,0 1280,56,256,64,1,1,0 2>/dev/null
  R3M_GG_DEBUG=$d python tools/conv_bench.py dgradbn 1280,56,64,64,3,1,1 2>/dev/null
done
} 2>&1 | tee gpurun_out/narrow_probe.txt
