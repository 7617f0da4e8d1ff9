"""-m gpu: the bf16-activation operators (`_dt` entry points with R3M_DT_BF16) through the C ABI.

Reference = the same op in fp64 on the bf16-ROUNDED inputs, so what is measured is the kernel (fp32 accumulation on the
bf16 MFMA) plus ONE output rounding to bf16 (relative step 2^-8 = 3.9e-3): tolerances are 2^-8 of the output range for bf16
outputs, and fp32-level for the fp32 outputs (weight gradients, BatchNorm partials and parameter gradients)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import nchw, nhwc, rel_err, rnd

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BF16 = 1
EPS_BF16 = 2.0 ** -8


def st():
    return torch.cuda.current_stream().cuda_stream


def q(x):
    """round to bf16, keep as fp32 (the value the device tensor holds)"""
    return x.to(torch.bfloat16).to(torch.float32)


# (N, H, Ci, Co, k, stride, pad): every ResNet-18/34/50 geometry behind the stem + ragged spatial sizes
CONV_CASES = [
    (2, 56, 64, 64, 1, 1, 0), (2, 56, 64, 64, 3, 1, 1), (2, 56, 64, 256, 1, 1, 0), (2, 56, 256, 64, 1, 1, 0),
    (2, 56, 256, 128, 1, 1, 0), (2, 56, 128, 128, 3, 2, 1), (2, 28, 128, 512, 1, 1, 0), (2, 56, 256, 512, 1, 2, 0),
    (2, 28, 512, 128, 1, 1, 0), (2, 28, 128, 128, 3, 1, 1), (2, 28, 256, 256, 3, 2, 1), (3, 14, 256, 1024, 1, 1, 0),
    (2, 28, 512, 1024, 1, 2, 0), (3, 14, 1024, 256, 1, 1, 0), (3, 14, 256, 256, 3, 1, 1), (3, 14, 512, 512, 3, 2, 1),
    (5, 7, 512, 2048, 1, 1, 0), (3, 14, 1024, 2048, 1, 2, 0), (5, 7, 2048, 512, 1, 1, 0), (5, 7, 512, 512, 3, 1, 1),
    (2, 56, 64, 128, 3, 2, 1), (2, 56, 64, 128, 1, 2, 0), (2, 28, 128, 256, 3, 2, 1), (3, 14, 256, 512, 3, 2, 1),
    (3, 9, 64, 64, 3, 1, 1), (1, 11, 64, 192, 3, 2, 1), (7, 5, 192, 64, 1, 1, 0),
    (84, 28, 128, 128, 3, 1, 1),      # M = 65 856 rows: the 256 x 128 halo tile (wide 3x3 stride-1 launches with M >= 65 536)
    (3, 8, 64, 128, 3, 1, 1), (5, 16, 128, 64, 3, 1, 1),   # image widths that are multiples of 8: the all-taps weight-gradient kernel
]


def _ids(c):
    return "N{}_H{}_{}to{}_k{}s{}p{}".format(*c)


@pytest.mark.parametrize("case", CONV_CASES, ids=_ids)
def test_conv_bf16_fwd_dgrad_wgrad(hip, case):
    N, H, Ci, Co, k, s, p = case
    x = q(rnd((N, Ci, H, H), 1))
    w = q(rnd((Co, Ci, k, k), 2, -0.2, 0.2))
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=p)
    Ho = y_ref.shape[2]
    dy = q(rnd(tuple(y_ref.shape), 3))
    y_ref.backward(dy.double())

    xd = nhwc(x).to(DEV).to(torch.bfloat16)
    w32 = w.permute(0, 2, 3, 1).contiguous().to(DEV)              # fp32 master, OHWI
    wd = torch.empty((Co, k, k, Ci), dtype=torch.bfloat16, device=DEV)
    assert hip.r3m_convert_bf16(w32.data_ptr(), wd.data_ptr(), w32.numel(), st()) == 0, hip.r3m_last_error()
    torch.testing.assert_close(wd.float().cpu(), w.permute(0, 2, 3, 1), rtol=0, atol=0)
    yd = torch.full((N, Ho, Ho, Co), float("nan"), dtype=torch.bfloat16, device=DEV)
    rows = hip.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
    stats = torch.zeros((rows, 2, Co), device=DEV)
    rc = hip.r3m_conv2d_fwd_dt(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, BF16, st())
    assert rc == 0, hip.r3m_last_error()
    yr = y_ref.detach()
    e_max, e_l2 = rel_err(nchw(yd.float().cpu()).numpy(), yr.numpy())
    assert e_max < EPS_BF16 and e_l2 < EPS_BF16 / 2, f"conv fwd bf16 max-rel {e_max} l2 {e_l2}"
    # BatchNorm partials come from the fp32 accumulators (before the bf16 rounding of y)
    np.testing.assert_allclose(stats[:, 0].double().sum(0).cpu().numpy(), yr.sum((0, 2, 3)).numpy(), rtol=1e-4,
                               atol=1e-3 * float(yr.abs().max()))
    np.testing.assert_allclose(stats[:, 1].double().sum(0).cpu().numpy(), (yr * yr).sum((0, 2, 3)).numpy(), rtol=1e-4)

    # dgrad
    dyd = nhwc(dy).to(DEV).to(torch.bfloat16)
    dxd = torch.full((N, H, H, Ci), float("nan"), dtype=torch.bfloat16, device=DEV)
    wsb = hip.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.r3m_conv2d_dgrad_dt(dyd.data_ptr(), w32.data_ptr(), dxd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, BF16, st())
    assert rc == 0, hip.r3m_last_error()
    e_max, e_l2 = rel_err(nchw(dxd.float().cpu()).numpy(), xr.grad.numpy())
    assert e_max < EPS_BF16 and e_l2 < EPS_BF16 / 2, f"conv dgrad bf16 max-rel {e_max} l2 {e_l2}"

    # wgrad: fp32 output (+ accumulate)
    dwd = torch.full((Co, k, k, Ci), float("nan"), device=DEV)
    wsb = hip.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, k, s, p, BF16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for acc in (0, 1):
        rc = hip.r3m_conv2d_wgrad_dt(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, acc,
                                     BF16, st())
        assert rc == 0, hip.r3m_last_error()
        e_max, _ = rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), (acc + 1) * wr.grad.numpy())
        assert e_max < 5e-5, f"conv wgrad bf16 (acc={acc}) max-rel {e_max}"


def test_conv_bf16_is_transpose_safe(hip):
    """one-hot rows against an ASYMMETRIC weight: exact in bf16, catches operand / C-layout transposes and k permutation bugs"""
    N, H, Ci, Co = 1, 8, 128, 128
    x = torch.zeros((N, H, H, Ci))
    for i in range(H * H):
        x.view(-1, Ci)[i, (3 * i) % Ci] = 1.0
    w = q(torch.arange(Co * Ci, dtype=torch.float32).view(Co, Ci, 1, 1) % 251)   # integers < 256: exact in bf16
    y_ref = F.conv2d(nchw(x), w)
    yd = torch.empty((N, H, H, Co), dtype=torch.bfloat16, device=DEV)
    xd, wd = x.to(DEV).to(torch.bfloat16), w.view(Co, 1, 1, Ci).contiguous().to(DEV).to(torch.bfloat16)
    assert hip.r3m_conv2d_fwd_dt(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), None, N, H, H, Ci, Co, 1, 1, 0, BF16, st()) == 0
    torch.testing.assert_close(nchw(yd.float().cpu()), y_ref, rtol=0, atol=0)
    # wgrad of the same: dW[co, ci] = sum_m dY[m, co] X[m, ci]   (exact small integers)
    dy = q(torch.arange(H * H * Co, dtype=torch.float32).view(N, H, H, Co) % 7)
    dw_ref = torch.einsum("mo,mi->oi", dy.view(-1, Co).double(), x.view(-1, Ci).double())
    dyd = dy.to(DEV).to(torch.bfloat16)
    dwd = torch.empty((Co, 1, 1, Ci), device=DEV)
    wsb = hip.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, 1, 1, 0, BF16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    assert hip.r3m_conv2d_wgrad_dt(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, 1, 1, 0, 0, BF16,
                                   st()) == 0, hip.r3m_last_error()
    torch.testing.assert_close(dwd.cpu().view(Co, Ci).double(), dw_ref, rtol=0, atol=0)


@pytest.mark.parametrize("Fr", [1, 3])
def test_stem_bf16_out(hip, Fr):
    """stem conv: fp32 frames / weights in (exact fp32 MFMA), bf16 activation out; wgrad reads a bf16 dY"""
    x = torch.floor(rnd((Fr, 3, 224, 224), 5, 0.0, 256.0)).clamp(0, 255)
    w = rnd((64, 3, 7, 7), 6, -0.1, 0.1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d((x / 255.0 - mean) / std, wr, stride=2, padding=3)
    dy = q(rnd(tuple(y_ref.shape), 7))
    y_ref.backward(dy)
    x_raw, wd = x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    xd = torch.empty((Fr, 224, 224, 3), device=DEV)
    assert hip.r3m_stem_prep(x_raw.data_ptr(), xd.data_ptr(), Fr, st()) == 0
    yd = torch.empty((Fr, 112, 112, 64), dtype=torch.bfloat16, device=DEV)
    stats = torch.zeros((Fr * 49, 2, 64), device=DEV)
    assert hip.r3m_stem_conv_fwd_dt(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), Fr, BF16, st()) == 0, hip.r3m_last_error()
    assert rel_err(nchw(yd.float().cpu()).numpy(), y_ref.detach().numpy())[0] < EPS_BF16
    yr = y_ref.detach().double()
    np.testing.assert_allclose(stats[:, 1].double().sum(0).cpu().numpy(), (yr * yr).sum((0, 2, 3)).numpy(), rtol=1e-4)
    dyd = nhwc(dy).to(DEV).to(torch.bfloat16)
    dwd = torch.empty((64, 7, 7, 3), device=DEV)
    wsb = hip.r3m_stem_conv_wgrad_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    assert hip.r3m_stem_conv_wgrad_dt(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, Fr, 0, BF16, st()) == 0
    assert rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), wr.grad.numpy())[0] < 5e-5


@pytest.mark.parametrize("rows,C", [(2 * 56 * 56, 64), (3 * 14 * 14, 1024), (5 * 49, 2048), (776, 256)])
@pytest.mark.parametrize("mode", ["plain", "identity", "downsample"])
def test_bn_bf16_fwd_bwd(hip, rows, C, mode):
    """BatchNorm(train) + [residual] + ReLU on bf16 activations, coefficients given (fp32); vs fp64 on the rounded inputs"""
    y = q(rnd((rows, C), 11, -2.0, 3.0))
    r = q(rnd((rows, C), 16, 0.0, 1.0))
    y2 = q(rnd((rows, C), 17, -1.0, 1.0))
    dz = q(rnd((rows, C), 20))
    gamma, beta = rnd((C,), 12, 0.5, 1.5), rnd((C,), 13, -0.3, 0.3)
    g2, b2 = rnd((C,), 18, 0.5, 1.5), rnd((C,), 19, -0.3, 0.3)

    def coef_of(yy, gg, bb):
        m = yy.double().mean(0)
        v = yy.double().var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(v + 1e-5)
        sc = gg.double() * inv
        return torch.stack([m, inv, sc, bb.double() - m * sc]).float()

    coef, coef2 = coef_of(y, gamma, beta), coef_of(y2, g2, b2)
    # reference with the SAME fp32 coefficients, fp64 arithmetic
    c = coef.double()
    t = y.double() * c[2] + c[3]
    if mode == "identity":
        t = t + r.double()
    elif mode == "downsample":
        c2 = coef2.double()
        t = t + (y2.double() * c2[2] + c2[3])
    z_ref = torch.relu(t)
    mask = (t > 0).double()
    g = dz.double() * mask
    yhat = (y.double() - c[0]) * c[1]
    db_ref, dg_ref = g.sum(0), (g * yhat).sum(0)
    dy_ref = c[2] * (g - g.mean(0) - yhat * (g * yhat).mean(0))

    bf = torch.bfloat16
    yd, zd = y.to(DEV).to(bf), torch.empty((rows, C), dtype=bf, device=DEV)
    coefd, coef2d = coef.to(DEV), coef2.to(DEV)
    bits = torch.zeros((rows * C + 31) // 32, dtype=torch.int32, device=DEV)
    rd, y2d = r.to(DEV).to(bf), y2.to(DEV).to(bf)
    if mode == "plain":
        rc = hip.r3m_bn_act_fwd_dt(yd.data_ptr(), coefd.data_ptr(), None, None, None, zd.data_ptr(), rows, C, 1, None, BF16, st())
    elif mode == "identity":
        rc = hip.r3m_bn_act_fwd_dt(yd.data_ptr(), coefd.data_ptr(), rd.data_ptr(), None, None, zd.data_ptr(), rows, C, 1, bits.data_ptr(), BF16, st())
    else:
        rc = hip.r3m_bn_act_fwd_dt(yd.data_ptr(), coefd.data_ptr(), None, y2d.data_ptr(), coef2d.data_ptr(), zd.data_ptr(), rows, C, 1,
                                   bits.data_ptr(), BF16, st())
    assert rc == 0, hip.r3m_last_error()
    assert rel_err(zd.float().cpu().numpy(), z_ref.numpy())[0] < EPS_BF16
    if mode != "plain":   # mask bits = [t > 0] evaluated in fp32 before the rounding; compare away from t ~ 0
        zb = bits.cpu().numpy().view(np.uint32)
        got = ((zb[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[: rows * C].reshape(rows, C)
        far = (t.abs() > 1e-4).numpy()
        assert np.array_equal(got[far], (t > 0).numpy()[far].astype(got.dtype))

    dzd = dz.to(DEV).to(bf)
    wsb = hip.r3m_bn_workspace_bytes(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dg, db, dyd = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty((rows, C), dtype=bf, device=DEV)
    rc = hip.r3m_bn_bwd_dt(dzd.data_ptr(), None, None if mode == "plain" else bits.data_ptr(), yd.data_ptr(), coefd.data_ptr(),
                           dg.data_ptr(), db.data_ptr(), dyd.data_ptr(), ws.data_ptr(), wsb, rows, C, 1, 0, BF16, st())
    assert rc == 0, hip.r3m_last_error()
    assert rel_err(dg.cpu().numpy(), dg_ref.numpy())[0] < 2e-4
    assert rel_err(db.cpu().numpy(), db_ref.numpy())[0] < 2e-4
    assert rel_err(dyd.float().cpu().numpy(), dy_ref.numpy())[0] < EPS_BF16


def test_pools_bf16(hip):
    bf = torch.bfloat16
    N, H, C = 2, 112, 64
    z = q(torch.relu(rnd((N, C, H, H), 41)))
    zr = z.clone().requires_grad_(True)
    p_ref = F.max_pool2d(zr, 3, 2, 1)
    Ho = p_ref.shape[2]
    dp = q(rnd(tuple(p_ref.shape), 42))
    p_ref.backward(dp)
    zd = nhwc(z).to(DEV).to(bf)
    pd = torch.empty((N, Ho, Ho, C), dtype=bf, device=DEV)
    am = torch.empty((N, Ho, Ho, C), dtype=torch.uint8, device=DEV)
    assert hip.r3m_maxpool_fwd_dt(zd.data_ptr(), pd.data_ptr(), am.data_ptr(), N, H, H, C, BF16, st()) == 0
    torch.testing.assert_close(nchw(pd.float().cpu()), p_ref.detach(), rtol=0, atol=0)
    dzd = torch.empty((N, H, H, C), dtype=bf, device=DEV)
    dpd = nhwc(dp).to(DEV).to(bf)
    assert hip.r3m_maxpool_bwd_dt(dpd.data_ptr(), am.data_ptr(), dzd.data_ptr(), N, H, H, C, BF16, st()) == 0
    mask = (z > 0).float()
    torch.testing.assert_close(nchw(dzd.float().cpu()) * mask, zr.grad * mask, rtol=2 * EPS_BF16, atol=2 * EPS_BF16)

    N, HW, C = 6, 49, 2048
    x = q(rnd((N, HW, C), 51))
    hd = torch.empty((N, C), device=DEV)
    xd = x.to(DEV).to(bf)
    assert hip.r3m_avgpool_fwd_dt(xd.data_ptr(), hd.data_ptr(), N, HW, C, BF16, st()) == 0
    assert rel_err(hd.cpu().numpy(), x.double().mean(1).numpy())[0] < 1e-6
    dh = rnd((N, C), 52)
    dxd = torch.empty((N, HW, C), dtype=bf, device=DEV)
    dhd = dh.to(DEV)
    assert hip.r3m_avgpool_bwd_dt(dhd.data_ptr(), dxd.data_ptr(), N, HW, C, BF16, st()) == 0
    torch.testing.assert_close(dxd.float().cpu(), q(dh / 49.0).unsqueeze(1).expand(N, HW, C), rtol=0, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
# engine level: the bf16 encoder against the (oracle-pinned) fp32 encoder on the same weights and frames
# ---------------------------------------------------------------------------------------------------------------------
def _report(line):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print(line)
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _grad_stats(ga, gb):
    """overall cosine, worst conv-weight cosine (tensors with > 0.1 % of the gradient energy), norm ratio |b|/|a|"""
    tot_a = sum(float(v.double().pow(2).sum()) for v in ga.values())
    dot = nb = 0.0
    worst, worst_name = 1.0, ""
    for k, a in ga.items():
        a, b = a.double().flatten(), gb[k].double().flatten()
        dot += float(a @ b)
        nb += float(b @ b)
        if ga[k].dim() == 4 and float(a @ a) > 1e-3 * tot_a:
            c = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
            if c < worst:
                worst, worst_name = c, k
    return dot / (tot_a * nb) ** 0.5, worst, worst_name, (nb / tot_a) ** 0.5


@pytest.mark.parametrize("size", [18, 34, 50])
def test_encoder_bf16_matches_emulation(hip, size):
    """The bf16 engine against oracle/bf16_emul.py (the oracle ResNet in float64 with a bf16 rounding at every tensor the
    engine stores in bf16), forward + backward, torchvision initialisation, 8 frames, with fixed (eval-mode, calibrated) and
    with batch BatchNorm statistics.

    Mixed precision cannot be bit-parity with the fp32 reference, and at initialisation on i.i.d.-noise frames the network
    is badly conditioned for ANY 2^-8 arithmetic: features are a large per-channel constant plus a small signal, BatchNorm
    subtracts the constant, and the relative rounding error of the stored tensor is amplified by |mean|/std at every layer.
    Measured here (MI355X, see profiles/r01_parity_report.txt): the exact bf16 arithmetic (the emulation) sits at gradient
    cosine 0.99 / 0.89 / 0.21 (ResNet-18/34/50, fixed statistics) and 0.91 / 0.72 / 0.11 (batch statistics) from the exact
    gradient, while the fp32 engine sits at 0.9998-1.0000. So the gates are relative: the HIP path must be CLOSER to the
    emulated bf16 arithmetic than that arithmetic is to exact (a kernel defect would add its own distance on top: embedding
    distance and 1 - cosine both <= 0.8x the format's with fixed statistics, <= 1x with batch statistics), the gradient norm must be preserved, and in the one well-conditioned
    case (ResNet-18, fixed statistics) the agreement must be tight in absolute terms. The per-operator tests above are the
    exact-to-one-rounding parity statement; test_train_steps_bf16_track_fp32 covers the training trajectory."""
    from oracle import bf16_emul, detgen, resnet_ref
    from r3m_amd import R3M
    N = 8 if size != 50 else 4            # the float64 CPU evaluations dominate this test's run time
    modes = (False, True) if size != 50 else (False,)
    torch.manual_seed(11)
    ref = getattr(resnet_ref, f"resnet{size}")().double()
    x = torch.from_numpy(detgen.frames("frames16", (16, 3, 224, 224)))[:N]
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64).view(1, 3, 1, 1)
    xn = (x.double() / 255.0 - mean) / std

    def exact(v):
        z = ref.maxpool(ref.relu(ref.bn1(ref.conv1(v))))
        return ref.layer4(ref.layer3(ref.layer2(ref.layer1(z)))).mean((2, 3))

    # calibrate the running statistics on this batch (momentum 1 -> running = batch statistics)
    bns = [m for m in ref.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    ref.train()
    with torch.no_grad():
        exact(xn)
    for m in bns:
        m.momentum = 0.1
    sd = {k: v.clone() for k, v in ref.state_dict().items() if not k.startswith("fc.")}

    def run_ref(fwd, training):
        ref.load_state_dict(sd, strict=False)
        ref.train(training)
        ref.zero_grad()
        h = fwd(xn)
        cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).double()
        (h * cw).sum().backward()
        return h.detach().clone(), {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}

    def run_hip(prec, training):
        m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision=prec)
        m.convnet.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()})
        m = m.to(DEV)
        m.train(training)
        h = m(x.to(DEV))
        cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).to(DEV)
        (h * cw).sum().backward()
        g = {k: p.grad.detach().cpu() for k, p in m.convnet.named_parameters()}
        assert all(torch.isfinite(v).all() for v in g.values())
        return h.detach().cpu().double(), g

    for training in modes:
        h_ex, g_ex = run_ref(exact, training)
        h_em, g_em = run_ref(lambda v: bf16_emul.forward_bf16(ref, v), training)
        h16, g16 = run_hip("bf16", training)
        h32, g32 = run_hip("fp32", training)
        d_fmt = float((h_em - h_ex).norm() / h_ex.norm())        # the format's distance from exact
        d_hip = float((h16 - h_em).norm() / h_em.norm())         # HIP's distance from the format
        d_32 = float((h32 - h_ex).norm() / h_ex.norm())
        c_fmt, w_fmt, _, _ = _grad_stats(g_ex, g_em)
        c_hip, w_hip, w_name, ratio = _grad_stats(g_em, g16)
        c_32, _, _, _ = _grad_stats(g_ex, g32)
        _report(f"r{size} bf16 {'batch-stat' if training else 'fixed-stat'} BN: h l2-rel emul~exact {d_fmt:.3e} hip16~emul {d_hip:.3e} "
                f"hip32~exact {d_32:.3e}; grad cosine emul~exact {c_fmt:.5f} (worst {w_fmt:.5f}) hip16~emul {c_hip:.5f} "
                f"(worst {w_hip:.5f} {w_name}) hip32~exact {c_32:.6f}; |g_hip16|/|g_emul| {ratio:.4f}")
        k = 1.0 if training else 0.8     # batch statistics: chaotic regime, "no farther than the format itself" is all one can ask
        assert d_hip <= k * d_fmt + 1e-3
        assert (1.0 - c_hip) <= k * (1.0 - c_fmt) + 1e-3 and (1.0 - w_hip) <= k * (1.0 - w_fmt) + 1e-3
        assert 0.9 <= ratio <= 1.1
        if size == 18 and not training:
            assert d_hip <= 6e-3 and c_hip >= 0.997 and w_hip >= 0.99 and 0.99 <= ratio <= 1.01


@pytest.mark.parametrize("size", [18, 34, 50])
def test_encoder_bf16_well_conditioned_absolute(hip, size):
    """a bf16 gate that CAN fail, for both BatchNorm modes at every size — in particular ResNet-50 with batch
    statistics, the mode that is trained and benched. The case is chosen so that the bf16 arithmetic itself is close to exact:
    low-frequency frames that differ from each other (oracle/detgen.smooth_frames) and a state with small residual branches
    (detgen.resnet_state_dict_small_residual: gamma of each block's last BatchNorm x 0.1). There the emulated format sits at
    gradient cosine >= 0.95 from the exact float64 gradient (tools/experiments/bf16_conditioning.py: ResNet-50 0.99998 fixed /
    0.997 batch statistics, against 0.21 / 0.10 for the plain initial state) — asserted below so the gate cannot go vacuous —
    and the HIP engine is then gated against the emulation in ABSOLUTE terms: embedding l2-rel <= 1e-2, overall gradient cosine
    >= 0.99, worst conv tensor >= 0.97, gradient norm within 3 %."""
    from oracle import bf16_emul, detgen, resnet_ref
    from r3m_amd import R3M
    N = 4 if size == 50 else 8
    ref = getattr(resnet_ref, f"resnet{size}")().double()
    shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items() if not k.startswith("fc.")]
    sd_np = detgen.resnet_state_dict_small_residual(shapes, size, 0.1)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    x = torch.from_numpy(detgen.smooth_frames("smooth", (N, 3, 224, 224), 7))
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64).view(1, 3, 1, 1)
    xn = (x.double() / 255.0 - mean) / std

    def exact(v):
        z = ref.maxpool(ref.relu(ref.bn1(ref.conv1(v))))
        return ref.layer4(ref.layer3(ref.layer2(ref.layer1(z)))).mean((2, 3))

    def run_ref(fwd, training):
        ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=False)
        ref.train(training)
        ref.zero_grad()
        h = fwd(xn)
        cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).double()
        (h * cw).sum().backward()
        return h.detach().clone(), {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}

    def run_hip(training):
        m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision="bf16")
        m.convnet.load_state_dict(sd)
        m = m.to(DEV)
        m.train(training)
        h = m(x.to(DEV))
        cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).to(DEV)
        (h * cw).sum().backward()
        g = {k: p.grad.detach().cpu() for k, p in m.convnet.named_parameters()}
        assert all(torch.isfinite(v).all() for v in g.values())
        return h.detach().cpu().double(), g

    for training in (False, True):
        h_ex, g_ex = run_ref(exact, training)
        h_em, g_em = run_ref(lambda v: bf16_emul.forward_bf16(ref, v), training)
        h16, g16 = run_hip(training)
        d_fmt = float((h_em - h_ex).norm() / h_ex.norm())
        d_hip = float((h16 - h_em).norm() / h_em.norm())
        c_fmt, w_fmt, _, _ = _grad_stats(g_ex, g_em)
        c_hip, w_hip, w_name, ratio = _grad_stats(g_em, g16)
        _report(f"r{size} bf16 WELL-CONDITIONED {'batch-stat' if training else 'fixed-stat'} BN: h l2-rel emul~exact {d_fmt:.3e} hip16~emul "
                f"{d_hip:.3e}; grad cosine emul~exact {c_fmt:.5f} (worst {w_fmt:.5f}) hip16~emul {c_hip:.5f} (worst {w_hip:.5f} {w_name}); "
                f"|g_hip16|/|g_emul| {ratio:.4f}")
        assert c_fmt >= 0.95, f"the case is not well conditioned any more (emulation vs exact cosine {c_fmt})"
        assert d_hip <= 1e-2, d_hip
        assert c_hip >= 0.99 and w_hip >= 0.97, (c_hip, w_hip, w_name)
        assert 0.97 <= ratio <= 1.03, ratio


AUTOCAST_DRAWS = [("w", "smooth"), ("wb", "smoothb"), ("wc", "smoothc")]     # (weight tag, frame tag) of oracle/detgen.py


def _last_bn_name(size):
    return "layer4.2.bn3" if size == 50 else ("layer4.2.bn2" if size == 34 else "layer4.1.bn2")


@pytest.mark.parametrize("size", [18, 50])
def test_encoder_bf16_against_torch_cpu_autocast_witness(hip, size):
    """an INDEPENDENT witness for the mixed-precision path. `oracle/bf16_emul.py` is this repository's own model of
    where the engine rounds; torch's own `torch.autocast("cpu", dtype=torch.bfloat16)` around the pinned oracle encoder (the graph
    behind /root/reference/r3m/models/models_r3m.py:96-99, fp32 master weights) is somebody else's. On the well-conditioned case
    (small residual branches, low-frequency frames — the case in which bf16 arithmetic is meaningful at all, see the test above),
    train-mode BatchNorm, three (weights, frames) draws, everything is measured against float64 truth of the same fp32 masters:
        err(X) = ||X - float64|| / ||float64||   for the embedding and seven named gradient tensors,
    and gated G8-style: the MEDIAN over the draws of err(HIP bf16) / err(torch autocast) must be <= 2 and every draw <= 4 — the HIP
    path may not be meaningfully worse than what `torch.autocast(bfloat16)` around the reference's encoder does on a CPU. The same
    three-way table is printed (and gated) for the emulation, so the checker of every other bf16 test is itself checked against torch.
    Where the two differ by construction: autocast keeps BatchNorm outputs, the residual sum and the pooled
    embedding in bf16 and takes batch statistics from bf16-rounded conv outputs; the engine takes them from the fp32 accumulators,
    keeps the embedding fp32, and rounds each stored tensor once."""
    from oracle import bf16_emul, detgen, resnet_ref
    from r3m_amd import R3M
    N = 4 if size == 50 else 8
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    lb = _last_bn_name(size)
    names = ["conv1.weight", "bn1.weight", "bn1.bias", lb + ".weight", lb + ".bias", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight"]

    def enc(m, v):
        z = m.maxpool(m.relu(m.bn1(m.conv1(v))))
        return m.layer4(m.layer3(m.layer2(m.layer1(z)))).mean((2, 3))

    def l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    rows = {k: {"hip": [], "emul": []} for k in ["h"] + names}
    for draw, (wtag, ftag) in enumerate(AUTOCAST_DRAWS):
        ref0 = getattr(resnet_ref, f"resnet{size}")()
        shapes = [(k, tuple(v.shape)) for k, v in ref0.state_dict().items() if not k.startswith("fc.")]
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict_small_residual(shapes, size, 0.1, tag=wtag).items()}
        x = torch.from_numpy(detgen.smooth_frames(ftag, (N, 3, 224, 224), 7))
        cw = None

        def run_cpu(dtype, mode):
            nonlocal cw
            m = getattr(resnet_ref, f"resnet{size}")().to(dtype)
            m.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, strict=False)
            m.train(True)
            xn = (x.to(dtype) / 255.0 - mean.to(dtype)) / std.to(dtype)
            if mode == "autocast":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    h = enc(m, xn)
                assert h.dtype == torch.bfloat16, "torch.autocast did not run the encoder in bfloat16"
                h = h.float()
            elif mode == "emul":
                h = bf16_emul.forward_bf16(m, xn)
            else:
                h = enc(m, xn)
            if cw is None:
                cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5))
            (h * cw.to(h.dtype)).sum().backward()
            return h.detach().double(), {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}

        h64, g64 = run_cpu(torch.float64, "exact")
        hac, gac = run_cpu(torch.float32, "autocast")
        hem, gem = run_cpu(torch.float64, "emul")
        m = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision="bf16")
        m.convnet.load_state_dict(sd)
        m = m.to(DEV)
        m.train(True)
        h = m(x.to(DEV))
        (h * cw.to(DEV)).sum().backward()
        ghip = {k: p.grad.detach().cpu().double() for k, p in m.convnet.named_parameters()}
        hhip = h.detach().cpu().double()
        assert torch.isfinite(hhip).all() and all(torch.isfinite(v).all() for v in ghip.values())
        for k in ["h"] + names:
            t64, tac, tem, thip = (h64, hac, hem, hhip) if k == "h" else (g64[k], gac[k], gem[k], ghip[k])
            e_ac, e_em, e_hip = l2(tac, t64), l2(tem, t64), l2(thip, t64)
            rows[k]["hip"].append(e_hip / max(e_ac, 1e-12))
            rows[k]["emul"].append(e_em / max(e_ac, 1e-12))
            _report(f"r{size} bf16 AUTOCAST WITNESS draw {draw} {k}: l2-rel vs float64: torch-autocast {e_ac:.3e}  hip-bf16 {e_hip:.3e}  "
                    f"emulation {e_em:.3e}  (hip/autocast {e_hip / max(e_ac, 1e-12):.2f}, emulation/autocast {e_em / max(e_ac, 1e-12):.2f})")
        del m
    fails = []
    for k, r in rows.items():
        for who in ("hip", "emul"):
            med = sorted(r[who])[1]
            _report(f"r{size} bf16 AUTOCAST WITNESS {k}: {who} / torch-autocast error ratios over the three draws "
                    f"{', '.join(f'{v:.2f}' for v in r[who])}  median {med:.2f}")
            if med > 2.0 or max(r[who]) > 4.0:
                fails.append((k, who, r[who]))
    assert not fails, fails


def test_train_steps_bf16_track_fp32(hip):
    """a few full Trainer.update steps with the bf16 encoder (TCN + LP loss, fused Adam on fp32 masters): finite metrics and
    the loss follows the fp32 run step by step (same data, same permutations)"""
    from r3m_amd import R3M
    from r3m_amd.parallel import make_network_wrapper
    from r3m_amd.trainer import Trainer
    losses = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(3)
        m = R3M("cuda", 1e-3, 1024, size=18, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0, precision=prec).to(DEV)
        net = make_network_wrapper(m)
        g = torch.Generator().manual_seed(7)
        frames = torch.randint(0, 256, (4, 5, 3, 224, 224), generator=g).float().to(DEV)
        tr = Trainer(eval_freq=10 ** 9)
        ls = []
        for i in range(6):
            torch.manual_seed(100 + i)          # same permutations in both runs
            met, _ = tr.update(net, (frames, [""] * 4), i)
            assert all(np.isfinite(v) for v in met.values()), met
            ls.append(met["full_loss"])
        losses[prec] = ls
    _report(f"bf16 train steps: fp32 {['%.4f' % v for v in losses['fp32']]} bf16 {['%.4f' % v for v in losses['bf16']]}")
    for a, b in zip(losses["fp32"], losses["bf16"]):      # same trajectory, step by step
        assert abs(a - b) <= 5e-2 * abs(a), (losses["fp32"], losses["bf16"])


@pytest.mark.parametrize("Fr", [1, 3])
def test_stem_on_bf16_mfma(hip, Fr):
    """the bf16 plan's stem: padded bf16 image of the normalised frames -> conv 7x7/2 forward (+ BatchNorm partials) and
    weight gradient on the bf16 MFMA, vs float64 on the bf16-rounded operands"""
    x = torch.floor(rnd((Fr, 3, 224, 224), 5, 0.0, 256.0)).clamp(0, 255)
    w = rnd((64, 3, 7, 7), 6, -0.1, 0.1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    xn = q((x / 255.0 - mean) / std)
    wr = q(w).double().requires_grad_(True)
    y_ref = F.conv2d(xn.double(), wr, stride=2, padding=3)
    dy = q(rnd(tuple(y_ref.shape), 7))
    y_ref.backward(dy.double())
    x_raw, wd = x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    nb = hip.r3m_stem_xn16_bytes(Fr)
    xn16 = torch.full((nb // 2,), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert hip.r3m_stem_prep_bf16(x_raw.data_ptr(), xn16.data_ptr(), Fr, st()) == 0, hip.r3m_last_error()
    img = xn16.view(Fr, 232, 704).float().cpu()
    torch.testing.assert_close(img[:, 3:227, 9:681].reshape(Fr, 224, 224, 3), xn.permute(0, 2, 3, 1), rtol=0, atol=0)
    assert float(img[:, :3].abs().max()) == 0 and float(img[:, 227:].abs().max()) == 0
    assert float(img[:, :, :9].abs().max()) == 0 and float(img[:, :, 681:].abs().max()) == 0
    yd = torch.full((Fr, 112, 112, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    stats = torch.zeros((Fr * 49, 2, 64), device=DEV)
    assert hip.r3m_stem_conv_fwd_bf16(xn16.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), Fr, st()) == 0, hip.r3m_last_error()
    yr = y_ref.detach()
    e_max, e_l2 = rel_err(nchw(yd.float().cpu()).numpy(), yr.numpy())
    assert e_max < EPS_BF16 and e_l2 < EPS_BF16 / 2, (e_max, e_l2)
    np.testing.assert_allclose(stats[:, 0].double().sum(0).cpu().numpy(), yr.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3 * float(yr.abs().max()))
    np.testing.assert_allclose(stats[:, 1].double().sum(0).cpu().numpy(), (yr * yr).sum((0, 2, 3)).numpy(), rtol=1e-4)
    dyd = nhwc(dy).to(DEV).to(torch.bfloat16)
    dwd = torch.full((64, 7, 7, 3), float("nan"), device=DEV)
    wsb = hip.r3m_stem_conv_wgrad_bf16_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for acc in (0, 1):
        assert hip.r3m_stem_conv_wgrad_bf16(xn16.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, Fr, acc, st()) == 0, hip.r3m_last_error()
        e_max, _ = rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), (acc + 1) * wr.grad.numpy())
        assert e_max < 5e-5, (acc, e_max)


# ---- round 6: the persistent kernel-row 3x3 kernel (csrc/conv_row16.hip) against the per-tile halo kernels it replaced ----------------
# Different K order (32-channel chunk -> kernel row -> tap vs 64-channel chunk -> tap): the fp32 accumulators differ by round-off, the
# stored bf16 elements by at most ONE ulp (of the element, floored at a quarter of the tensor's rms so that a cancelled sum is not
# judged against its own tiny value), and each kernel is within half an ulp of float64 truth. Cases: both tile widths, M not a
# multiple of 512, a tile that starts in front of the tensor, W = 56 / 30 (the widest windows), odd frame sizes.
ROW16_CASES = [(5, 28, 128, 128), (3, 14, 256, 256), (9, 7, 512, 512), (40, 14, 128, 256), (2, 28, 64, 128), (1, 30, 128, 128),
               (2, 56, 64, 64), (3, 28, 128, 64), (1, 9, 64, 64), (5, 56, 64, 64), (23, 7, 64, 384)]


def _ulp_of(ref):
    mag = torch.maximum(ref.abs(), 0.25 * ref.pow(2).mean().sqrt())
    return torch.exp2(torch.floor(torch.log2(mag)) - 7)


@pytest.mark.parametrize("case", ROW16_CASES, ids=lambda c: "N{}_H{}_{}to{}".format(*c))
def test_conv3x3_kernel_row_kernel_vs_halo_kernel_and_float64(hip, case):
    N, H, Ci, Co = case
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn((N, H, H, Ci), device=DEV, generator=g).bfloat16()
    w = (torch.randn((Co, 3, 3, Ci), device=DEV, generator=g) * (1.5 / (9 * Ci) ** 0.5)).bfloat16()
    dy = torch.randn((N, H, H, Co), device=DEV, generator=g).bfloat16()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).contiguous()
    dref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).contiguous()
    rows = hip.r3m_conv2d_stats_rows(N, H, H, Co, 3, 1, 1)
    wsb = hip.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, 3)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    wf = w.float()
    buf = (__import__("ctypes").c_int * 4)()
    assert hip.r3m_debug_conv_route(N, H, H, Ci, Co, 3, 1, 1, 0, 1, 0, 1, buf, 4) == 1 and buf[0] == 32, "not on the kernel-row route"
    out = {}
    old = hip.r3m_debug_set_conv3x3_bf16(1)
    try:
        for mode in (1, 0):
            hip.r3m_debug_set_conv3x3_bf16(mode)
            y = torch.full((N, H, H, Co), float("nan"), dtype=torch.bfloat16, device=DEV)
            stats = torch.full((rows, 2, Co), float("nan"), device=DEV)
            dx = torch.full((N, H, H, Ci), float("nan"), dtype=torch.bfloat16, device=DEV)
            assert hip.r3m_conv2d_fwd_dt(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, 3, 1, 1, BF16, st()) == 0, hip.r3m_last_error()
            assert hip.r3m_conv2d_dgrad_dt(dy.data_ptr(), wf.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, 3, 1, 1, BF16, st()) == 0, hip.r3m_last_error()
            torch.cuda.synchronize()
            out[mode] = (y, stats, dx)
    finally:
        hip.r3m_debug_set_conv3x3_bf16(old)
    for name, new, base, truth in (("y", out[1][0], out[0][0], ref), ("dx", out[1][2], out[0][2], dref)):
        ulp = _ulp_of(truth)
        d = ((new.double() - base.double()).abs() / ulp).max().item()
        e_new = ((new.double() - truth).abs() / ulp).max().item()
        e_old = ((base.double() - truth).abs() / ulp).max().item()
        assert torch.isfinite(new.float()).all()
        assert d <= 1.0, f"{name}: kernel-row vs halo kernel {d} ulp"
        assert e_new <= 0.51 and e_old <= 0.51, f"{name}: vs float64 {e_new} / {e_old} ulp"
    # BatchNorm partials: the same rows (one per 128 result rows; per 256 for 64-wide outputs) from both kernels, fp32-level agreement
    s1, s0 = out[1][1].double(), out[0][1].double()
    assert torch.isfinite(s1).all() and s1.shape == s0.shape
    assert (s1 - s0).abs().max().item() <= 3e-6 * s0.abs().max().item()
    M = N * H * H
    SR = 128 if Co % 128 == 0 else 256
    s_ref = torch.stack([ref.reshape(M, Co)[r:r + SR].sum(0) for r in range(0, M, SR)])
    q_ref = torch.stack([(ref.reshape(M, Co)[r:r + SR] ** 2).sum(0) for r in range(0, M, SR)])
    assert ((s1[:, 0] - s_ref).abs().max() / s_ref.abs().max()).item() < 2e-6
    assert ((s1[:, 1] - q_ref).abs().max() / q_ref.abs().max()).item() < 2e-6


def test_conv3x3_kernel_row_kernel_rows_do_not_depend_on_the_frame_count(hip):
    """Plans of different frame counts must produce the same rows (tests/test_gpu_fullsize.py compares a 2560-frame plan with
    320-frame ones): the kernel-row kernel's accumulation order is a property of the layer, not of M."""
    H, Ci, Co = 14, 256, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((700, H, H, Ci), device=DEV, generator=g).bfloat16()
    w = (torch.randn((Co, 3, 3, Ci), device=DEV, generator=g) * 0.03).bfloat16()
    ys = []
    for N in (700, 3):
        y = torch.empty((N, H, H, Co), dtype=torch.bfloat16, device=DEV)
        stats = torch.empty((hip.r3m_conv2d_stats_rows(N, H, H, Co, 3, 1, 1), 2, Co), device=DEV)
        assert hip.r3m_conv2d_fwd_dt(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, 3, 1, 1, BF16, st()) == 0, hip.r3m_last_error()
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0][:3].view(torch.int16), ys[1].view(torch.int16))
