#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
SH="1280,14,256,1024,1,1,0 1280,28,128,512,1,1,0 1280,14,1024,256,1,1,0 1280,14,256,256,3,1,1 1280,7,512,2048,1,1,0 1280,7,512,512,3,1,1"
python tools/conv_bench.py fwd16 1280,14,256,256,3,1,1 1280,14,256,256,3,1,1 > /dev/null   # clock ramp
for E in "A=0" "R3M_BF16_BIG=-1" "R3M_BF16_BIG=-1 R3M_GG_DEBUG=1" "R3M_BF16_BIG=-1 R3M_GG_DEBUG=2" "R3M_BF16_BIG=-1 R3M_GG_DEBUG=3" "R3M_BF16_BK=64" "R3M_BF16_BK=32"; do echo "== $E"; env $E python tools/conv_bench.py fwd16 $SH; done
