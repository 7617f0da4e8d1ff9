#!/usr/bin/env python
"""Search for the kink-free golden case: a detgen state whose float64 last-block pre-activations z = bn(y) +
identity have NO element within `margin` of zero, so that no fp32 forward can decide a last-block ReLU differently from float64
and the gradient gate needs no flip accounting. The state is oracle.detgen.resnet_state_dict_no_kink: the plain generator under
a new tag with the LAST BatchNorm's bias shifted up by `shift` (its inputs are ~N(0,1)-normalised, so the mass of z near zero
falls like the normal tail). usage: find_nokink.py [sizes...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import detgen, r3m_ref  # noqa: E402

torch.set_num_threads(8)


def min_abs_z(size, tag, shift, frames_tag, F=8):
    m = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    sd = detgen.resnet_state_dict_no_kink(shapes, size, tag=tag, shift=shift)
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.double()
    m.train()
    net = m.convnet
    blk = net.layer4[-1]
    last = blk.bn3 if size == 50 else blk.bn2
    keep = {}
    hooks = [last.register_forward_hook(lambda mod, i, o: keep.__setitem__("bn_out", o.detach())),
             blk.register_forward_pre_hook(lambda mod, i: keep.__setitem__("idn", i[0].detach()))]
    x = torch.from_numpy(detgen.frames(frames_tag, (F, 3, 224, 224))).double()
    with torch.no_grad():
        net(m.normlayer(x / 255.0))
    for h in hooks:
        h.remove()
    z = (keep["bn_out"] + keep["idn"]).flatten()
    a = z.abs()
    return float(a.min()), int((a < 2e-4).sum()), int((a < 1e-3).sum()), float((z < 0).double().mean()), float(a.mean())


if __name__ == "__main__":
    sizes = [int(s) for s in sys.argv[1:]] or [18, 34, 50]
    for size in sizes:
        for shift in (3.0, 4.0, 5.0):
            for tag in ("nk", "nk2", "nk3"):
                mn, n2, n10, neg, mean = min_abs_z(size, tag, shift, "frames8nk")
                print(f"r{size} tag {tag} shift {shift}: min|z| {mn:.3e}  #|z|<2e-4 {n2}  #|z|<1e-3 {n10}  frac(z<0) {neg:.4f}  mean|z| {mean:.2f}", flush=True)
                if n10 == 0:
                    break
