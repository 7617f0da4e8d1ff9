#!/usr/bin/env python
"""Micro-benchmark of single conv launches through the C ABI (GPU only). usage: conv_bench.py [fwd|wgrad] N,H,Ci,Co,k,s,p ..."""
import sys
import torch
sys.path.insert(0, ".")
from r3m_amd import _lib

L = _lib.lib()
mode = sys.argv[1]
for spec in sys.argv[2:]:
    N, H, Ci, Co, k, s, p = [int(v) for v in spec.split(",")]
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn((N, H, H, Ci), device="cuda")
    w = torch.randn((Co, k, k, Ci), device="cuda") * 0.05
    y = torch.empty((N, Ho, Ho, Co), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * N * Ho * Ho * Co * Ci * k * k
    if mode == "fwd":
        rows = L.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
        stats = torch.empty((rows, 2, Co), device="cuda")
        fn = lambda: L.r3m_conv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, st)
    else:
        dy = torch.randn_like(y)
        dw = torch.empty_like(w)
        wsb = L.r3m_conv2d_wgrad_workspace_bytes(N, H, H, Ci, Co, k, s, p)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        fn = lambda: L.r3m_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 0, st)
    for _ in range(3):
        assert fn() == 0, L.r3m_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    M = N * Ho * Ho
    print(f"{mode} {spec:28s} M={M:9d} ms={ms:8.3f} TF/s={flops/ms/1e9:7.1f}  tiles128={-(-M//128)*max(1,Co//128)} rounds={-(-M//128)*max(1,Co//128)/768:.3f}")
