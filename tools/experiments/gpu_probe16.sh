#!/bin/bash
# timing probes of the bf16 GEMM on the expanding / contracting 1x1 and a 3x3 shape: 0 full, 1 no epilogue, 2 no DMA after the prologue, 3 both
mkdir -p gpurun_out; export TMPDIR=/tmp
SH="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,56,64,256,1,1,0 1280,14,1024,256,1,1,0 1280,14,256,256,3,1,1 1280,56,64,64,3,1,1"
python tools/conv_bench.py fwd16 1280,14,256,256,3,1,1 1280,14,256,256,3,1,1 > /dev/null   # clock ramp
for D in 0 1 2 3; do echo "== R3M_GG_DEBUG=$D"; R3M_GG_DEBUG=$D python tools/conv_bench.py fwd16 $SH; done
echo "== fp32"; python tools/conv_bench.py fwd $SH
