#!/usr/bin/env python
"""GPU: why does a conv launch run 10-25 % slower inside the step than in a micro-benchmark? Three conditions
for one fp32 forward launch (EPI_STATS), same process:
    burst      40 back-to-back launches (what tools/conv_bench.py measures)
    sustained  ~3 s of back-to-back launches (the board's sustained-load clock, operands partly resident in the 256 MB Infinity Cache)
    evicted    every launch preceded by a 1 GiB streaming fill (what a BatchNorm pass leaves behind: operands out of every cache, dirty
               lines draining); time = (fill + launch) - (fill alone), both sustained
usage: sustained_probe.py N,H,Ci,Co,k,s,p ..."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import _lib

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
junk = torch.empty(1 << 28, device="cuda", dtype=torch.float32)   # 1 GiB


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for spec in sys.argv[1:]:
    N, H, Ci, Co, k, s, p = [int(v) for v in spec.split(",")]
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn((N, H, H, Ci), device="cuda")
    w = torch.randn((Co, k, k, Ci), device="cuda") * 0.05
    y = torch.empty((N, Ho, Ho, Co), device="cuda")
    rows = L.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
    stats = torch.empty((2 * rows + 2, 2, Co), device="cuda")
    conv = lambda: L.r3m_conv2d_fwd_dt(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, 0, st)
    fill = lambda: junk.fill_(1.0)
    both = lambda: (junk.fill_(1.0), conv())
    flops = 2.0 * N * Ho * Ho * Co * Ci * k * k
    for _ in range(3):
        assert conv() == 0, L.r3m_last_error()
    burst = timed(conv, 40)
    n_sus = max(200, int(3000.0 / burst))
    sus = timed(conv, n_sus)
    burst2 = timed(conv, 40)
    t_fill = timed(fill, 300)
    t_both = timed(both, 300)
    ev = t_both - t_fill
    tf = lambda ms: flops / ms / 1e9
    print(f"{spec:26s} burst {burst:.3f} ms ({tf(burst):6.1f} TF/s)  sustained x{n_sus} {sus:.3f} ms ({tf(sus):6.1f})  burst again {burst2:.3f} ({tf(burst2):6.1f})  "
          f"after a 1 GiB fill {ev:.3f} ms ({tf(ev):6.1f}; fill alone {t_fill:.3f} ms)", flush=True)
