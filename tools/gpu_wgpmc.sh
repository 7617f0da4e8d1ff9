#!/bin/bash
# HBM read bytes of single wgrad launches (bf16 and fp32), with and without the XCD-aware block order
export TMPDIR=/tmp; mkdir -p gpurun_out
REPO=$(pwd)
L="1280,14,256,256,3,1,1 1280,56,64,64,3,1,1 1280,28,128,512,1,1,0"
for x in 0 1; do for m in wgrad16 wgrad; do
  rm -rf /tmp/pw
  (cd /tmp && R3M_WG_XCD=$x timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pw -o p -- python $REPO/tools/conv_bench.py $m $L > /dev/null 2>&1)
  db=$(find /tmp/pw -name "*.db" | head -1)
  echo "== R3M_WG_XCD=$x $m (FETCH_SIZE KiB, x2 for wide reads; 13 launches per shape, shapes summed per kernel)"
  python tools/rocpd_pmc.py "$db" | grep -i "wgrad" | head -6
done; done
