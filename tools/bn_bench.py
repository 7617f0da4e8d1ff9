#!/usr/bin/env python
"""Micro-benchmark of the BatchNorm passes through the C ABI (GPU only): rows,C ... for fp32 and bf16; prints ms and TB/s."""
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
    rows, C = [int(v) for v in spec.split(",")]
    for dt, tdt, eb in ((0, torch.float32, 4), (1, torch.bfloat16, 2)):
        y = torch.randn((rows, C), device="cuda").to(tdt)
        r = torch.randn((rows, C), device="cuda").to(tdt)
        dz = torch.randn((rows, C), device="cuda").to(tdt)
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        coef = torch.rand((4, C), device="cuda") + 0.5
        bits = torch.zeros((rows * C + 31) // 32, dtype=torch.int32, device="cuda")
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        wsb = L.r3m_bn_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        n = rows * C
        t0 = timeit(lambda: L.r3m_bn_act_fwd_dt(y.data_ptr(), coef.data_ptr(), None, None, None, z.data_ptr(), rows, C, 1, None, dt, st))
        t1 = timeit(lambda: L.r3m_bn_act_fwd_dt(y.data_ptr(), coef.data_ptr(), r.data_ptr(), None, None, z.data_ptr(), rows, C, 1, bits.data_ptr(), dt, st))
        t2 = timeit(lambda: L.r3m_bn_bwd_dt(dz.data_ptr(), None, bits.data_ptr(), y.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                            dy.data_ptr(), ws.data_ptr(), wsb, rows, C, 1, 0, dt, st))
        print(f"{'bf16' if dt else 'fp32'} rows={rows:9d} C={C:5d}  act_fwd {t0:7.3f} ms {2*n*eb/t0/1e9:5.2f} TB/s | act_fwd+res+bits {t1:7.3f} ms "
              f"{3*n*eb/t1/1e9:5.2f} TB/s | bwd(reduce+apply) {t2:7.3f} ms {5*n*eb/t2/1e9:5.2f} TB/s")
