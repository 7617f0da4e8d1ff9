#!/usr/bin/env python
"""GPU: does EPI_BNRED (BatchNorm-backward partials from the producing dgrad's epilogue: one read of dOut and of y saved per BatchNorm)
pay on the bf16 plans now? Rounds 2-5 measured it neutral-to-slower there and left it off; round 6 found the bf16 step power-bound
(profiles/r06_row16_dvfs_*.txt), where bytes not moved are energy not spent. Same process, same model, the per-plan switch
r3m_resnet_set_fused_bn_reduce flipped between interleaved legs of `STEPS` training steps each.
usage: bnred16_ab.py c2|c4 [legs] [steps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import R3M, _lib
from r3m_amd.parallel import make_network_wrapper
from r3m_amd.trainer import Trainer

cfg = sys.argv[1]
legs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = _lib.lib()
dev = "cuda:0"
torch.manual_seed(1)
if cfg == "c2":
    size, B, lw, aug = 50, 256, 1.0, "none"
else:
    size, B, lw, aug = 34, 512, 0.0, "rctraj"
model = R3M("cuda", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=lw, tcnweight=1.0, l2dist=True, bs=B, precision="bf16").to(dev)
net = make_network_wrapper(model)
g = torch.Generator(device=dev).manual_seed(1234)
if aug == "none":
    frames = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()
    get = lambda: frames
else:
    from r3m_amd import augment
    raw = torch.randint(0, 256, (B, 5, 3, 256, 256), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    bg = torch.Generator().manual_seed(99)
    get = lambda: augment.random_resized_crop(raw, per_clip=True, generator=bg, fused=True)
langs = [""] * B
if lw > 0:
    langs = torch.randn((B, 768), generator=torch.Generator(device=dev).manual_seed(4321), device=dev) * 0.3
tr = Trainer(eval_freq=10 ** 9)
for i in range(6):
    tr.update(net, (get(), langs), i)
torch.cuda.synchronize()


def set_all(on):
    n = 0
    for sl in model.convnet._ring:
        for plan in sl.plans.values():
            L.r3m_resnet_set_fused_bn_reduce(plan, on)
            n += 1
    return n


res = {0: [], 1: []}
for leg in range(legs):
    for on in (0, 1):
        assert set_all(on) >= 1
        for i in range(3):
            tr.update(net, (get(), langs), i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.update(net, (get(), langs), i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[on].append(ms)
        print(f"{cfg} leg {leg} fused_bn_reduce={on}: {ms:.3f} ms/step", flush=True)
set_all(0)
print(f"{cfg} mean: off {sum(res[0]) / legs:.3f} ms  on {sum(res[1]) / legs:.3f} ms")
