#!/usr/bin/env python
"""GPU: the inference forward (r3m_resnet_forward training = 2: BatchNorm + residual + ReLU in the convolutions' stores) against the
unfused eval sequence (r3m_debug_set_fused_inference(0): conv, then a stand-alone bn_act_fwd pass), same process, interleaved legs.
usage: inference_ab.py [size] [frames] [legs]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import R3M, _lib

size = int(sys.argv[1]) if len(sys.argv) > 1 else 50
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
legs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = _lib.lib()
for prec in ("fp32", "bf16"):
    torch.manual_seed(1)
    m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision=prec).to("cuda:0").eval()
    x = torch.randint(0, 256, (frames, 3, 224, 224), device="cuda:0").float()
    res = {0: [], 1: []}
    with torch.no_grad():
        for _ in range(5):
            m(x)
        for leg in range(legs):
            for on in (0, 1):
                L.r3m_debug_set_fused_inference(on)
                for _ in range(3):
                    m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    m(x)
                torch.cuda.synchronize()
                res[on].append((time.perf_counter() - t0) / 10 * 1e3)
    L.r3m_debug_set_fused_inference(1)
    u, f = sum(res[0]) / legs, sum(res[1]) / legs
    print(f"ResNet-{size} {prec} {frames} frames: unfused eval {u:.2f} ms ({frames / u * 1e3:.0f} frames/s)  fused inference {f:.2f} ms "
          f"({frames / f * 1e3:.0f} frames/s)  {100 * (u - f) / u:.1f} % less   legs unfused {[round(v, 2) for v in res[0]]} fused {[round(v, 2) for v in res[1]]}")
    del m
    torch.cuda.empty_cache()
