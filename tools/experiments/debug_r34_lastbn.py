"""CPU (float64 oracle): is the 1.3e-3 error of the HIP d(gamma) of ResNet-34's last BatchNorm a ReLU-kink flip?
Lists the elements of the last block's pre-ReLU sum closest to 0 and what flipping each would do to the gamma/beta gradient."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import detgen, r3m_ref  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 34
torch.set_num_threads(8)
m = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0)
shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
m = m.double()
x = torch.from_numpy(detgen.frames("frames8", (8, 3, 224, 224))).double()
m.train()
net = m.convnet
blk = net.layer4[-1]
keep = {}
last = blk.bn3 if size == 50 else blk.bn2
h1 = last.register_forward_hook(lambda mod, i, o: keep.__setitem__("bn_out", o.detach()))
h0 = last.register_forward_pre_hook(lambda mod, i: keep.__setitem__("bn_in", i[0].detach()))
hb = blk.register_forward_pre_hook(lambda mod, i: keep.__setitem__("idn", i[0].detach()))
h = net(m.normlayer(x / 255.0))
cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).double()
(h * cw).sum().backward()
z = keep["bn_out"] + keep["idn"]                      # pre-ReLU sum of the last block [8, C, 7, 7]
y = keep["bn_in"]
mu = y.mean((0, 2, 3), keepdim=True)
var = y.var((0, 2, 3), unbiased=False, keepdim=True)
yhat = (y - mu) / torch.sqrt(var + 1e-5)
dz = (cw / 49.0).view(8, -1, 1, 1).expand_as(z)
dg = (dz * (z > 0) * yhat).sum((0, 2, 3))
db = (dz * (z > 0)).sum((0, 2, 3))
print("check vs autograd: dgamma", float((dg - last.weight.grad).abs().max()), "dbeta", float((db - last.bias.grad).abs().max()))
az = z.abs().flatten()
idx = torch.argsort(az)[:12]
print(f"|dgamma| = {float(dg.norm()):.4e}  |dbeta| = {float(db.norm()):.4e}   scale of z: {float(z.abs().mean()):.3f}")
for i in idx.tolist():
    n, c, p, q = np.unravel_index(i, z.shape)
    eff_g = float(abs(dz[n, c, p, q] * yhat[n, c, p, q])) / float(dg.norm())
    eff_b = float(abs(dz[n, c, p, q])) / float(db.norm())
    print(f"  z[{n},{c},{p},{q}] = {float(z.flatten()[i]):+.3e}   a flip changes |dgamma| by {eff_g:.3e} (l2-rel), |dbeta| by {eff_b:.3e}")
