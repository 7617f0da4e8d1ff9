#!/usr/bin/env python
"""GPU box: WHERE does the fp32 HIP path's gradient round-off exceed PyTorch-CPU fp32's on the BasicBlock nets, and is that a property
of this build or of any GPU fp32 convolution path? tools/grad_noise_bisect.py showed that no schedule
switch of the engine changes the figure; this tool adds witnesses and a per-tensor-group breakdown. For one kink-free state and K
redrawn loss-weight vectors, against the float64 gradients of the pinned oracle (oracle/r3m_ref.R3MRef on this box's CPU):
    cpu-fp32        the oracle in float32 on the CPU, all threads                   (the "reference" of the G8 gate)
    cpu-fp32-1t     the same with ONE thread: another blocking / summation order of the same library
    rocm-eager-fp32 the oracle module moved to the GPU and run by torch's own ROCm kernels (MIOpen convolutions, ATen BatchNorm) —
                    test infrastructure only: a GPU fp32 path nobody here wrote
    hip             this build (R3M, fp32)
Printed: root-mean-square over the samples of the l2-rel error of the gradient, for the whole vector and per tensor group (stem conv,
conv weights of layer1..4, BatchNorm weights, BatchNorm biases), and every witness's ratio to cpu-fp32.
usage: grad_noise_witness.py SIZE [K] [draw]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detgen, r3m_ref
from r3m_amd import R3M

size = int(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
draw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
DEV = "cuda:0"
NT = max(1, min(64, len(os.sched_getaffinity(0)) // 2))
torch.set_num_threads(NT)
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
tag, shift, ftag = detgen.NOKINK_STATES[size][draw]
m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0).to(DEV)
shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict_no_kink(shapes, size, tag=tag, shift=shift).items()}
m.convnet.load_state_dict(sd)
m.train()
x = torch.from_numpy(detgen.frames(ftag, (8, 3, 224, 224)))


def oracle_grads(dtype, cw, device="cpu"):
    ref = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0).to(dtype).to(device)
    ref.convnet.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, strict=False)
    ref.train()
    obs = x.to(dtype).to(device) / 255.0
    h = ref.convnet(ref.normlayer(obs))
    (h * cw.to(dtype).to(device)).sum().backward()
    return {k: p.grad.detach().double().cpu() for k, p in ref.convnet.named_parameters() if p.grad is not None}


def group_of(name):
    if name == "conv1.weight":
        return "stem conv"
    if name.endswith("weight") and ("conv" in name or "downsample.0" in name):
        return "conv " + name.split(".")[0]
    return "bn weight" if name.endswith("weight") else "bn bias"


WIT = ["cpu-fp32", "cpu-fp32-1t", "rocm-eager-fp32", "hip"]
err = {w: {} for w in WIT}
for k in range(K):
    cw = torch.from_numpy(detgen.uniform(f"cwb{k}", (8, m.outdim), 0.5, 1.5))
    g64 = oracle_grads(torch.float64, cw)
    names = list(g64.keys())
    groups = sorted({group_of(n) for n in names})
    res = {"cpu-fp32": oracle_grads(torch.float32, cw)}
    torch.set_num_threads(1)
    res["cpu-fp32-1t"] = oracle_grads(torch.float32, cw)
    torch.set_num_threads(NT)
    try:
        res["rocm-eager-fp32"] = oracle_grads(torch.float32, cw, DEV)
    except Exception as e:  # MIOpen unavailable on the box: say so once, carry on with the others
        if k == 0:
            print("rocm-eager-fp32 unavailable:", repr(e)[:200], flush=True)
    m.encoder_opt.zero_grad()
    h = m(x.to(DEV))
    (h * cw.to(DEV)).sum().backward()
    res["hip"] = {n: p.grad.detach().cpu().double() for n, p in m.convnet.named_parameters()}
    for w, g in res.items():
        for grp in ["all"] + groups:
            sel = [n for n in names if grp == "all" or group_of(n) == grp]
            a = torch.cat([g[n].double().flatten() for n in sel])
            b = torch.cat([g64[n].flatten() for n in sel])
            err[w].setdefault(grp, []).append(float((a - b).norm() / b.norm()))
    print(f"sample {k}: " + "  ".join(f"{w} {err[w]['all'][-1]:.3e}" for w in WIT if "all" in err[w]), flush=True)
rms = lambda a: float(np.sqrt(np.mean(np.square(a))))
print(f"r{size} draw {draw} ({tag}, {ftag}), {K} loss-weight samples; rms over the samples of the l2-rel gradient error vs float64 (ratio to cpu-fp32):")
for grp in ["all"] + groups:
    base = rms(err["cpu-fp32"][grp])
    line = f"  {grp:12s}"
    for w in WIT:
        if grp in err[w]:
            line += f"  {w} {rms(err[w][grp]):.3e} ({rms(err[w][grp]) / base:.2f})"
    print(line)
