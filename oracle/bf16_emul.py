"""Checker for the mixed-precision (bf16 activation) encoder: the oracle ResNet (oracle/resnet_ref.py, the torchvision
restatement behind /root/reference/r3m/models/models_r3m.py:44-52,99) evaluated in float64 with a round-to-bfloat16 inserted at
exactly the points where the HIP engine stores a bf16 tensor. TEST INFRASTRUCTURE ONLY (tests/ and smoke); never imported
by the product.

The reference itself is fp32-only; "bf16" is BASELINE.json configs[2]/[4]. What this emulation pins down is that the HIP
bf16 path computes  fp32-master weights -> bf16 operands -> exact products, wide accumulation -> ONE rounding per stored
tensor, so that its distance from the fp32 path is the distance of the arithmetic, not of a kernel defect:

  forward   the normalised input frames, every conv output, BatchNorm(+residual)+ReLU output and the max-pool output are stored
            bf16; conv weights (the stem's included) are rounded to bf16 for the GEMM; BatchNorm statistics, coefficients and the
            pooled embedding stay wide.
  backward  every activation gradient (BatchNorm input gradient, conv input gradient incl. the residual join, max-pool and
            avg-pool input gradient) is stored bf16; weight and BatchNorm-parameter gradients stay wide.
Known, deliberately ignored differences (O(2^-9 / sqrt(count)) on statistics, one extra rounding on the downsample join):
the engine takes BatchNorm statistics from the un-rounded accumulators, and adds the downsample branch's input gradient to
an already rounded tensor.
"""
import torch
import torch.nn.functional as F


def _round(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _QAct(torch.autograd.Function):
    """stored activation: value rounded on the way forward, its gradient rounded on the way back"""

    @staticmethod
    def forward(ctx, x):
        return _round(x)

    @staticmethod
    def backward(ctx, g):
        return _round(g)


class _QWeight(torch.autograd.Function):
    """GEMM operand copy of a master weight: rounded forward, gradient passed through wide"""

    @staticmethod
    def forward(ctx, w):
        return _round(w)

    @staticmethod
    def backward(ctx, g):
        return g


def _conv(m, x):
    return _QAct.apply(F.conv2d(x, _QWeight.apply(m.weight), None, m.stride, m.padding))


def _bn(m, x):
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps)


def _block(b, x):
    convs = [(b.conv1, b.bn1), (b.conv2, b.bn2)] + ([(b.conv3, b.bn3)] if hasattr(b, "conv3") else [])
    out = x
    for i, (c, n) in enumerate(convs):
        out = _bn(n, _conv(c, out))
        if i < len(convs) - 1:
            out = _QAct.apply(torch.relu(out))
    idn = x if b.downsample is None else _bn(b.downsample[1], _conv(b.downsample[0], x))
    return _QAct.apply(torch.relu(out + idn))


def forward_bf16(resnet, x_normalized):
    """resnet: oracle.resnet_ref.ResNet (any float dtype, fc ignored); x_normalized: (x/255 - mean)/std, NCHW.
    Returns the pooled embedding [N, D] (wide)."""
    m = resnet
    # stem: the engine keeps the normalised frames as a bf16 image and runs conv1 on the bf16 MFMA like every other conv
    y = _conv(m.conv1, _round(x_normalized))
    z = _QAct.apply(torch.relu(_bn(m.bn1, y)))
    z = _QAct.apply(F.max_pool2d(z, 3, 2, 1))
    for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
        for b in layer:
            z = _block(b, z)
    return z.mean((2, 3))
