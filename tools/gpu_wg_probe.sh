#!/bin/bash
# Where does the time of the fp32 3x3 weight gradient (kernel-row blocks) and of the 3x3 window kernel go? Probe build
# (tools/build_ab.sh none probes). R3M_WG_DEBUG bits: 1 no DMA, 2 no X pieces, 4 no dY pieces; R3M_GG_DEBUG (window kernel): 8 no
# window DMA, 16 no weight DMA, 4 no epilogue. Results are WRONG with any bit set.
mkdir -p gpurun_out; export TMPDIR=/tmp
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
S3="1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 1280,56,64,64,3,1,1"
S1="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0"
{
for d in 0 1 2 4; do
  echo "== wgrad R3M_WG_DEBUG=$d"
  R3M_WG_DEBUG=$d python tools/conv_bench.py wgrad $S3 $S1 2>/dev/null
done
echo "== wgrad per-tap blocks (R3M_WG_ROWS=0)"
R3M_WG_ROWS=0 python tools/conv_bench.py wgrad $S3 2>/dev/null
for d in 0 8 16 24 4 28; do
  echo "== window kernel fwd R3M_GG_DEBUG=$d"
  R3M_GG_DEBUG=$d python tools/conv_bench.py fwd 1280,28,128,128,3,1,1 1280,14,256,256,3,1,1 1280,7,512,512,3,1,1 2>/dev/null
done
} 2>&1 | tee gpurun_out/wg_probe.txt
