#!/usr/bin/env python
"""Micro-benchmark of single conv launches through the C ABI (GPU only).
env: CONV_BENCH_DATA=randn|zeros|ones (operand values), CONV_BENCH_REPS (timed launches, default 10)
usage: conv_bench.py [fwd|wgrad|dgradbn|dgradbnres|fwd16|wgrad16|...] N,H,Ci,Co,k,s,p ...   (the *16 modes run the bf16 kernels)
dgradbn: dgrad with the EPI_BNRED epilogue (mask recomputed from y); dgradbnres: + mask bits + masked residual-gradient join"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import _lib

L = _lib.lib()
if os.environ.get("PW16_MODE") and hasattr(L, "r3m_debug_set_pw16"):
    L.r3m_debug_set_pw16(int(os.environ["PW16_MODE"]))     # 0 per-tile kernels, 1 persistent pointwise / gather, 3 + window form
mode = sys.argv[1]
bf16 = mode.endswith("16")
if bf16:
    mode = mode[:-2]
for spec in sys.argv[2:]:
    N, H, Ci, Co, k, s, p = [int(v) for v in spec.split(",")]
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn((N, H, H, Ci), device="cuda")
    w = torch.randn((Co, k, k, Ci), device="cuda") * 0.05
    data = os.environ.get("CONV_BENCH_DATA", "randn")      # zeros / ones: the same instruction stream on operands that do not toggle the multipliers
    if data != "randn":
        x.fill_(0.0 if data == "zeros" else 1.0)
        w.fill_(0.0 if data == "zeros" else 1.0)
    y = torch.empty((N, Ho, Ho, Co), device="cuda")
    if bf16:
        x, w, y = x.bfloat16(), w.bfloat16(), y.bfloat16()
    dt = 1 if bf16 else 0
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * N * Ho * Ho * Co * Ci * k * k
    if mode == "fwd":
        rows = L.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
        stats = torch.empty((2 * rows + 2, 2, Co), device="cuda")   # (room for experiments with finer statistics rows)
        fn = lambda: L.r3m_conv2d_fwd_dt(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, dt, st)
    elif mode in ("dgradbn", "dgradbnres"):
        dy = torch.randn_like(y) if data == "randn" else torch.full_like(y, 0.0 if data == "zeros" else 1.0)
        dx = torch.empty_like(x)
        by = torch.randn_like(x)
        res = torch.randn_like(x) if mode == "dgradbnres" else None
        nbits = x.numel() // 32
        rbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (nbits,), device="cuda", dtype=torch.int32) if mode == "dgradbnres" else None
        ybits = torch.randint(-2 ** 31, 2 ** 31 - 1, (nbits,), device="cuda", dtype=torch.int32) if mode == "dgradbnres" else None
        sc, sh, mu = torch.rand(Ci, device="cuda") + 0.5, torch.rand(Ci, device="cuda") - 0.5, torch.rand(Ci, device="cuda")
        rows = L.r3m_conv2d_dgrad_bnred_rows(N, H, H, s)
        part = torch.empty((rows, 2, Ci), device="cuda")
        wsb = L.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        wf = w.float()
        P = lambda t: None if t is None else t.data_ptr()
        fn = lambda: L.r3m_conv2d_dgrad_bnred_dt(dy.data_ptr(), wf.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p,
                                                 P(res), P(rbits), by.data_ptr(), P(ybits), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                                                 part.data_ptr(), dt, st)
    else:
        dy = torch.randn_like(y) if data == "randn" else torch.full_like(y, 0.0 if data == "zeros" else 1.0)
        dw = torch.empty(w.shape, device="cuda")
        wsb = L.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, k, s, p, dt)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        fn = lambda: L.r3m_conv2d_wgrad_dt(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 0, dt, st)
    for _ in range(3):
        assert fn() == 0, L.r3m_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("CONV_BENCH_REPS", "10"))
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    M = N * Ho * Ho
    print(f"{mode}{'16' if bf16 else ''} {spec:28s} M={M:9d} ms={ms:8.3f} TF/s={flops/ms/1e9:7.1f}  tiles128={-(-M//128)*max(1,Co//128)} rounds={-(-M//128)*max(1,Co//128)/768:.3f}")
