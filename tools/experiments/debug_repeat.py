"""GPU: is the encoder forward bit-repeatable within one process? (eval mode, F = 15)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import detgen
from r3m_amd import R3M, augment
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 15
def build():
    m = R3M("cuda", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0, precision=prec)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    return m.to("cuda:0")
g = torch.Generator().manual_seed(21)
x = torch.randint(0, 256, (F, 3, 224, 224), generator=g, dtype=torch.uint8).cuda().float()
m = build()
for training in (False, True):
    m.train(training)
    with torch.no_grad():
        a = m(x).clone(); b = m(x).clone()
    print(prec, "training", training, "no_grad twice equal:", torch.equal(a, b), float((a - b).abs().max()))
    h = m(x); c = h.detach().clone()
    h.sum().backward()
    h2 = m(x); d = h2.detach().clone()
    print(prec, "training", training, "fwd / (bwd) / fwd equal:", torch.equal(c, d), float((c - d).abs().max()), " vs no_grad:", torch.equal(a, c))
    g1 = m.convnet.flat_grads().clone()
    m.encoder_opt.zero_grad(); h2.sum().backward()
    print(prec, "training", training, "grads equal:", torch.equal(g1, m.convnet.flat_grads()))
m2 = build(); m2.eval()
m.eval()
with torch.no_grad():
    print(prec, "fresh model vs used model (eval):", torch.equal(m2(x), m(x)))
