#!/usr/bin/env python
"""GPU: N inference forwards (load_r3m-style eval forward under no_grad) of one encoder, for rocprofv3 --kernel-trace:
usage: inference_loop.py fp32|bf16 [size] [frames] [calls]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import R3M

prec = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 50
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.manual_seed(1)
m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision=prec).to("cuda:0").eval()
x = torch.randint(0, 256, (frames, 3, 224, 224), device="cuda:0").float()
with torch.no_grad():
    m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        m(x)
    torch.cuda.synchronize()
print(f"ResNet-{size} {prec} {frames} frames: {(time.perf_counter() - t0) / calls * 1e3:.2f} ms per call")
