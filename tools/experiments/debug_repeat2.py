"""GPU: training-mode backward repeatability — which parameter gradients differ between two backward passes of the same forward?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import detgen
from r3m_amd import R3M
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 15
size = int(sys.argv[3]) if len(sys.argv) > 3 else 18
m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision=prec)
shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
m = m.to("cuda:0")
g = torch.Generator().manual_seed(21)
x = torch.randint(0, 256, (F, 3, 224, 224), generator=g, dtype=torch.uint8).cuda().float()
m.train()
res = []
for rep in range(3):
    h = m(x)
    m.encoder_opt.zero_grad()
    h.sum().backward()
    torch.cuda.synchronize()
    res.append({k: p.grad.clone() for k, p in m.convnet.named_parameters()})
for a, b, tag in ((0, 1, "run0 vs run1"), (1, 2, "run1 vs run2")):
    bad = [(k, float((res[a][k] - res[b][k]).abs().max()), float(res[a][k].abs().max())) for k in res[a] if not torch.equal(res[a][k], res[b][k])]
    print(prec, f"F={F} r{size}", tag, ":", len(bad), "of", len(res[a]), "tensors differ")
    for k, d, s in bad[:12]:
        print(f"    {k:36s} max|d| {d:.3e}  max|g| {s:.3e}")
