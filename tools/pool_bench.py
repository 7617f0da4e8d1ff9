#!/usr/bin/env python
"""Micro-benchmark of the fused stem tail (BatchNorm + ReLU + MaxPool forward; MaxPool + ReLU + BatchNorm backward) through the
C ABI (GPU only). usage: pool_bench.py N[,H[,C]] ...   (R3M_HIP_LIB selects an A/B library). Prints ms and TB/s of the algorithmic
bytes: forward = Y + P + argmax; backward = two reads of (Y + dP + argmax) + one write of dY."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import _lib

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=10):
    for _ in range(3):
        assert fn() == 0, L.r3m_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    N, H, C = v[0], (v[1] if len(v) > 1 else 112), (v[2] if len(v) > 2 else 64)
    Ho = (H + 2 - 3) // 2 + 1
    rows = N * H * H
    for dt, tdt, eb in ((0, torch.float32, 4), (1, torch.bfloat16, 2)):
        y = torch.randn((rows, C), device="cuda").to(tdt)
        coef = torch.stack([torch.zeros(C), torch.ones(C), torch.rand(C) + 0.5, torch.rand(C) * 0.2 - 0.1]).cuda().contiguous()
        p = torch.empty((N, Ho, Ho, C), dtype=tdt, device="cuda")
        am = torch.empty((N, Ho, Ho, C), dtype=torch.uint8, device="cuda")
        dp = torch.randn((N, Ho, Ho, C), device="cuda").to(tdt)
        dy = torch.empty_like(y)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        wsb = L.r3m_bn_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        tf = timeit(lambda: L.r3m_bn_relu_maxpool_fwd_dt(y.data_ptr(), coef.data_ptr(), p.data_ptr(), am.data_ptr(), N, H, H, C, dt, st))
        tb = timeit(lambda: L.r3m_bn_maxpool_bwd_dt(dp.data_ptr(), am.data_ptr(), y.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                    dy.data_ptr(), ws.data_ptr(), wsb, N, H, H, C, 1, 0, dt, st))
        yb, pb, ab = rows * C * eb, N * Ho * Ho * C * eb, N * Ho * Ho * C
        print(f"N={N} H={H} C={C} {'fp32' if dt == 0 else 'bf16'}: fwd {tf:7.3f} ms {(yb + pb + ab) / tf / 1e9:5.2f} TB/s   "
              f"bwd (reduce + apply) {tb:7.3f} ms {(2 * (yb + pb + ab) + yb) / tb / 1e9:5.2f} TB/s")
        del y, p, am, dp, dy, ws
