#!/bin/bash
# Where does a read-modify-write expanding dgrad (pw_gemm_kernel<128,128,2,2,EPI_BNRED|EPI_MASKED_ADD,YBITS>: the dgrad of a
# bottleneck block's first 1x1 convolution, which also adds the masked residual gradient and emits the consumer BatchNorm's backward
# partial sums) spend its time? Probe build (tools/build_ab.sh none probes). R3M_GG_DEBUG bits: 1 no result stores, 2 no statistics,
# 4 no epilogue at all, 8 no operand DMA, 16 no epilogue-operand prefetch (add0 / y / mask words). WRONG results with any bit.
# Beside each: the plain dgrad of the same shape (mode dgrad... = conv_bench "dgradbn" without residual) for the cost of the fusion.
mkdir -p gpurun_out; export TMPDIR=/tmp
S="1280,56,256,64,1,1,0 1280,28,512,128,1,1,0 1280,14,1024,256,1,1,0 1280,7,2048,512,1,1,0"
export R3M_HIP_LIB=$PWD/r3m_amd/lib/libr3m_hip_probes.so
for d in 0 1 16 17 4 8 24; do
  echo "== R3M_GG_DEBUG=$d"
  R3M_GG_DEBUG=$d python tools/conv_bench.py dgradbnres $S 2>/dev/null | cut -c1-80
done 2>&1 | tee gpurun_out/r05_rmw_probe.txt
echo "== shipped build, dgradbnres / dgradbn" | tee -a gpurun_out/r05_rmw_probe.txt
unset R3M_HIP_LIB
python tools/conv_bench.py dgradbnres $S 2>/dev/null | cut -c1-80 | tee -a gpurun_out/r05_rmw_probe.txt
python tools/conv_bench.py dgradbn $S 2>/dev/null | cut -c1-80 | tee -a gpurun_out/r05_rmw_probe.txt
