#!/usr/bin/env python
"""Bitwise A/B of the fp32 3x3 weight gradient: shared-window kernel (wgrad_win.hip) against the kernel-row form (conv.hip).
Probe build only (R3M_WG_WIN switches the dispatch):  wgrad_win_check.py save out.pt  /  wgrad_win_check.py cmp a.pt b.pt"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [(37, 28, 128, 128), (70, 14, 256, 256), (130, 7, 512, 512), (9, 56, 64, 64), (5, 9, 64, 128), (3, 13, 128, 64)]
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        d = (a[k] - b[k]).abs().max().item()
        print(k, "bit-identical" if same else f"DIFFERENT max abs {d:.3e} (max |dW| {a[k].abs().max().item():.3e})")
    sys.exit(0)
from r3m_amd import _lib
L = _lib.lib()
out = {}
g = torch.Generator(device="cuda").manual_seed(7)
for (N, H, Ci, Co) in CASES:
    x = torch.randn((N, H, H, Ci), device="cuda", generator=g)
    dy = torch.randn((N, H, H, Co), device="cuda", generator=g)
    dw = torch.empty((Co, 3, 3, Ci), device="cuda")
    wsb = L.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, 3, 1, 1, 0)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.r3m_conv2d_wgrad_dt(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, 3, 1, 1, 0, 0, st) == 0, L.r3m_last_error()
    torch.cuda.synchronize()
    out[f"N{N}_H{H}_{Ci}to{Co}"] = dw.cpu()
torch.save(out, sys.argv[2])
