#!/usr/bin/env python
"""What does another stream's long-running kernel cost the step? (GPU only.) A stand-in for an RCCL collective overlapped with
backward: `--blocks` workgroups that each hold `--lds` bytes of LDS sit on a side stream for the whole measurement
(r3m_debug_occupy), while the ResNet-50 fp32 step runs on the main stream — once with the persistent 1x1 kernel's tiles assigned
statically, once through its per-XCD tile queues (conv_pw.hip). usage: occupy_ab.py [--blocks 64] [--lds 98304] [--steps 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import R3M, _lib  # noqa: E402
from r3m_amd.parallel import SingleDevice  # noqa: E402
from r3m_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=64)
ap.add_argument("--lds", type=int, default=96 * 1024)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--clips", type=int, default=256)
a = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = R3M("cuda", 1e-4, 1024, size=50, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0, bs=a.clips).to(dev)
net = SingleDevice(m)
g = torch.Generator(device=dev).manual_seed(1234)
frames = torch.randint(0, 256, (a.clips, 5, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()
tr = Trainer(10 ** 9)
side = torch.cuda.Stream(device=dev)


def run(dynamic, occupied):
    L.r3m_debug_set_dynamic_tiles(1 if dynamic else 0)
    for i in range(3):
        tr.update(net, (frames, [""] * a.clips), i)
    torch.cuda.synchronize()
    if occupied:
        _lib.check(L.r3m_debug_occupy(a.blocks, a.lds, (a.steps * 0.45 + 0.3) * 1e3, side.cuda_stream), "debug_occupy")
        time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.steps):
        tr.update(net, (frames, [""] * a.clips), 3 + i)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    torch.cuda.synchronize()
    return ms


for occupied in (False, True):
    for dynamic in (False, True, False, True):
        ms = run(dynamic, occupied)
        print(f"{'occupied: %d blocks x %d KB LDS on a side stream' % (a.blocks, a.lds // 1024) if occupied else 'GPU to itself':52s} "
              f"tiles {'queues ' if dynamic else 'static '}: {ms:8.2f} ms / step", flush=True)
L.r3m_debug_set_dynamic_tiles(1)
