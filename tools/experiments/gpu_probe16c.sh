#!/bin/bash
# what-if probes for a halo-tile 3x3: stage the A tile of 3 (probe 8) or 1 (probe 16) of the 9 taps only (wrong results)
mkdir -p gpurun_out; export TMPDIR=/tmp
SH="1280,14,256,256,3,1,1 1280,28,128,128,3,1,1 1280,56,64,64,3,1,1 1280,7,512,512,3,1,1"
python tools/conv_bench.py fwd16 1280,14,256,256,3,1,1 1280,14,256,256,3,1,1 > /dev/null
for D in 0 8 16 2; do echo "== R3M_GG_DEBUG=$D"; R3M_GG_DEBUG=$D python tools/conv_bench.py fwd16 $SH; done
