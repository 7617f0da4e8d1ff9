mkdir -p gpurun_out; export TMPDIR=/tmp
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
{
for d in 0 32; do
  echo "== R3M_GG_DEBUG=$d (32: DMA pieces of a K step in one burst before its MFMAs)"
  R3M_GG_DEBUG=$d python tools/conv_bench.py fwd 1280,56,256,64,1,1,0 1280,56,64,256,1,1,0 1280,28,128,512,1,1,0 1280,28,512,128,1,1,0 1280,14,256,1024,1,1,0 1280,14,1024,256,1,1,0 1280,7,512,2048,1,1,0 1280,56,64,64,3,1,1 1280,28,256,512,1,2,0 2>/dev/null
  R3M_GG_DEBUG=$d python tools/conv_bench.py dgradbn 1280,56,64,256,1,1,0 1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 2>/dev/null
done
} 2>&1 | tee gpurun_out/cluster_probe.txt
