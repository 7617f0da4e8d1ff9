#!/usr/bin/env python
"""GPU: same-process A/B of one process-wide library switch (an `r3m_debug_set_*` entry point of include/r3m_hip.h) on whole training
steps of a BASELINE config: same model, same arena, interleaved legs of `steps` steps each with the switch off / on.
usage: switch_ab.py <switch> c1|c2|c4 [legs] [steps] [on-value]      e.g.  switch_ab.py conv3x3_bf16 c4 3 10   (on-value: what "on" passes, default 1)
(c1 = configs[1] ResNet-50 fp32 1280 frames; c2 = configs[2] ResNet-50 bf16 + language head; c4 = configs[4] ResNet-34 bf16 rctraj)"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import R3M, _lib
from r3m_amd.parallel import make_network_wrapper
from r3m_amd.trainer import Trainer

switch, cfg = sys.argv[1], sys.argv[2]
legs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
on_value = int(sys.argv[5]) if len(sys.argv) > 5 else 1
L = _lib.lib()
setter = getattr(L, "r3m_debug_set_" + switch)
dev = "cuda:0"
torch.manual_seed(1)
size, B, lw, aug, prec = {"c1": (50, 256, 0.0, "none", "fp32"), "c2": (50, 256, 1.0, "none", "bf16"),
                          "c4": (34, 512, 0.0, "rctraj", "bf16")}[cfg]
model = R3M("cuda", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=lw, tcnweight=1.0, l2dist=True, bs=B,
            precision=prec).to(dev)
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
for i in range(5):
    tr.update(net, (get(), langs), i)
torch.cuda.synchronize()
res = {0: [], 1: []}
default = setter(1)
for leg in range(legs):
    for on in (0, 1):
        setter(on_value if on else 0)
        for i in range(2):
            tr.update(net, (get(), langs), i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.update(net, (get(), langs), i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[on].append(ms)
        print(f"{cfg} leg {leg} {switch}={on}: {ms:.3f} ms/step", flush=True)
setter(default)
print(f"{cfg} {switch} mean: off {sum(res[0]) / legs:.3f} ms  on {sum(res[1]) / legs:.3f} ms  (default {default})")
