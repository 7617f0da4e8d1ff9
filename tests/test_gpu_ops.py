"""-m gpu: every HIP operator through the C ABI vs the PyTorch-CPU fp32 op the reference would run (fp32 tolerance written
at each assert). Shapes cover every distinct conv geometry of ResNet-18/34/50 at small batch, plus
ragged sizes that exercise the tile-edge masks."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import nchw, nhwc, rel_err, rnd

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def st():
    return torch.cuda.current_stream().cuda_stream


# (N, H, Ci, Co, k, stride, pad)
CONV_CASES = [
    (2, 56, 64, 64, 1, 1, 0), (2, 56, 64, 64, 3, 1, 1), (2, 56, 64, 256, 1, 1, 0), (2, 56, 256, 64, 1, 1, 0),
    (2, 56, 256, 128, 1, 1, 0), (2, 56, 128, 128, 3, 2, 1), (2, 28, 128, 512, 1, 1, 0), (2, 56, 256, 512, 1, 2, 0),
    (2, 28, 512, 128, 1, 1, 0), (2, 28, 128, 128, 3, 1, 1), (2, 28, 256, 256, 3, 2, 1), (3, 14, 256, 1024, 1, 1, 0),
    (2, 28, 512, 1024, 1, 2, 0), (3, 14, 1024, 256, 1, 1, 0), (3, 14, 256, 256, 3, 1, 1), (3, 14, 512, 512, 3, 2, 1),
    (5, 7, 512, 2048, 1, 1, 0), (3, 14, 1024, 2048, 1, 2, 0), (5, 7, 2048, 512, 1, 1, 0), (5, 7, 512, 512, 3, 1, 1),
    (2, 56, 64, 128, 3, 2, 1), (2, 56, 64, 128, 1, 2, 0), (2, 28, 128, 256, 3, 2, 1), (3, 14, 256, 512, 3, 2, 1),
    (3, 9, 64, 64, 3, 1, 1), (1, 11, 96, 192, 3, 2, 1), (7, 5, 160, 64, 1, 1, 0), (2, 13, 32, 64, 3, 2, 1),
    # persistent 1x1 kernel (conv_pw.hip): more tiles than resident workers (several tiles per worker, deferred epilogues),
    # partial last row tile, one / two / four K-step pairs per tile
    (24, 56, 64, 256, 1, 1, 0), (23, 56, 64, 128, 1, 1, 0), (21, 56, 128, 256, 1, 1, 0), (85, 28, 256, 128, 1, 1, 0),
    (24, 56, 256, 64, 1, 1, 0), (23, 56, 128, 64, 1, 1, 0),     # ... and its eight-wave 256 x 64 tile (64-channel outputs)
    # ... its gather form (3x3 / strided forward, stride-1 dgrad) and the strided-output form (parity classes of a stride-2 dgrad)
    (45, 28, 128, 128, 3, 1, 1), (90, 56, 128, 128, 3, 2, 1), (91, 56, 64, 64, 3, 2, 1), (90, 56, 128, 256, 1, 2, 0),
    # shared-window 3x3 weight gradient (wgrad_win.hip): image rows shorter than a K step's 32 rows by a lot (seven row segments per
    # step), longer than it (steps without a row start), and a width of 4
    (7, 5, 64, 64, 3, 1, 1), (3, 33, 64, 128, 3, 1, 1), (5, 4, 128, 128, 3, 1, 1),
]


def _ids(c):
    return "N{}_H{}_{}to{}_k{}s{}p{}".format(*c)


@pytest.mark.parametrize("case", CONV_CASES, ids=_ids)
def test_conv_fwd_dgrad_wgrad(hip, case):
    N, H, Ci, Co, k, s, p = case
    x = rnd((N, Ci, H, H), 1)
    w = rnd((Co, Ci, k, k), 2, -0.2, 0.2)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=p)
    Ho = y_ref.shape[2]
    dy = rnd(tuple(y_ref.shape), 3)
    y_ref.backward(dy)

    xd = nhwc(x).to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)       # OHWI
    yd = torch.empty((N, Ho, Ho, Co), device=DEV)
    rows = hip.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
    stats = torch.zeros((rows, 2, Co), device=DEV)
    rc = hip.r3m_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, st())
    assert rc == 0, hip.r3m_last_error()
    e_max, e_l2 = rel_err(nchw(yd.cpu()).numpy(), y_ref.detach().numpy())
    assert e_max < 2e-5, f"conv fwd max-rel {e_max}"
    # BatchNorm statistic partials: sum / sum of squares over rows, per output channel
    ssum = stats[:, 0].double().sum(0).cpu().numpy()
    ssq = stats[:, 1].double().sum(0).cpu().numpy()
    yr = y_ref.detach().double()
    np.testing.assert_allclose(ssum, yr.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3 * float(yr.abs().max()))
    np.testing.assert_allclose(ssq, (yr * yr).sum((0, 2, 3)).numpy(), rtol=1e-4)

    # dgrad
    dyd = nhwc(dy).to(DEV)
    dxd = torch.full((N, H, H, Ci), float("nan"), device=DEV)
    wsb = hip.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.r3m_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, st())
    assert rc == 0, hip.r3m_last_error()
    dx = nchw(dxd.cpu())
    e_max, _ = rel_err(dx.numpy(), xr.grad.numpy())   # (strided 1x1: odd pixels must come back as exact zeros)
    assert e_max < 2e-5, f"conv dgrad max-rel {e_max}"

    # wgrad (+ accumulate)
    dwd = torch.empty((Co, k, k, Ci), device=DEV)
    wsb = hip.r3m_conv2d_wgrad_workspace_bytes(N, H, H, Ci, Co, k, s, p)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = hip.r3m_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 0, st())
    assert rc == 0, hip.r3m_last_error()
    dw = dwd.cpu().permute(0, 3, 1, 2)
    e_max, _ = rel_err(dw.numpy(), wr.grad.numpy())
    assert e_max < 5e-5, f"conv wgrad max-rel {e_max}"
    rc = hip.r3m_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 1, st())
    assert rc == 0
    e_max, _ = rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), 2 * wr.grad.numpy())
    assert e_max < 5e-5, f"conv wgrad accumulate max-rel {e_max}"


def test_conv_is_transpose_safe(hip):
    """A = identity-like input against an ASYMMETRIC weight catches row/col swaps in the MFMA C layout."""
    N, H, Ci, Co = 1, 8, 64, 128
    x = torch.zeros((N, H, H, Ci))
    for i in range(H * H):
        x.view(-1, Ci)[i, i % Ci] = 1.0 + i
    w = torch.arange(Co * Ci, dtype=torch.float32).view(Co, Ci, 1, 1) / 100.0
    y_ref = F.conv2d(nchw(x), w)
    yd = torch.empty((N, H, H, Co), device=DEV)
    xd, wd = x.to(DEV), w.view(Co, 1, 1, Ci).contiguous().to(DEV)   # keep alive: launches are asynchronous
    rc = hip.r3m_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), None, N, H, H, Ci, Co, 1, 1, 0, st())
    assert rc == 0
    torch.testing.assert_close(nchw(yd.cpu()), y_ref, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("Fr", [1, 3])
def test_stem_direct_fwd_wgrad(hip, Fr):
    """The engine's stem: /255 -> Normalize -> conv 7x7/2 p3 straight from NCHW frames, forward (+BN partials) and wgrad."""
    x = torch.floor(rnd((Fr, 3, 224, 224), 5, 0.0, 256.0)).clamp(0, 255)
    w = rnd((64, 3, 7, 7), 6, -0.1, 0.1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d((x / 255.0 - mean) / std, wr, stride=2, padding=3)
    dy = rnd(tuple(y_ref.shape), 7)
    y_ref.backward(dy)
    x_raw, wd = x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    xd = torch.empty((Fr, 224, 224, 3), device=DEV)
    assert hip.r3m_stem_prep(x_raw.data_ptr(), xd.data_ptr(), Fr, st()) == 0
    torch.testing.assert_close(xd.cpu(), ((x / 255.0 - mean) / std).permute(0, 2, 3, 1), rtol=0, atol=0)   # same IEEE arithmetic
    yd = torch.empty((Fr, 112, 112, 64), device=DEV)
    stats = torch.zeros((Fr * 49, 2, 64), device=DEV)
    assert hip.r3m_stem_conv_fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), Fr, st()) == 0, hip.r3m_last_error()
    assert rel_err(nchw(yd.cpu()).numpy(), y_ref.detach().numpy())[0] < 2e-5
    yr = y_ref.detach().double()
    np.testing.assert_allclose(stats[:, 0].double().sum(0).cpu().numpy(), yr.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3 * float(yr.abs().max()))
    np.testing.assert_allclose(stats[:, 1].double().sum(0).cpu().numpy(), (yr * yr).sum((0, 2, 3)).numpy(), rtol=1e-4)
    dyd = nhwc(dy).to(DEV)
    dwd = torch.empty((64, 7, 7, 3), device=DEV)
    wsb = hip.r3m_stem_conv_wgrad_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    for acc in (0, 1):
        assert hip.r3m_stem_conv_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, Fr, acc, st()) == 0
        e_max, _ = rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), (acc + 1) * wr.grad.numpy())
        assert e_max < 5e-5, (acc, e_max)


@pytest.mark.parametrize("rows,C", [(2 * 56 * 56, 64), (3 * 14 * 14, 1024), (5 * 49, 2048), (777, 256), (33, 128)])
@pytest.mark.parametrize("mode", ["plain", "identity", "downsample"])
def test_bn_train_fwd_bwd(hip, rows, C, mode):
    """BatchNorm2d(train) + [residual] + ReLU, forward and backward, vs torch CPU (F.batch_norm + autograd)."""
    y = rnd((rows, C), 11, -2.0, 3.0)
    gamma, beta = rnd((C,), 12, 0.5, 1.5), rnd((C,), 13, -0.3, 0.3)
    rm, rv = rnd((C,), 14, -0.1, 0.1), rnd((C,), 15, 0.5, 1.5)
    r = rnd((rows, C), 16, 0.0, 1.0)
    y2 = rnd((rows, C), 17, -1.0, 1.0)
    g2, b2 = rnd((C,), 18, 0.5, 1.5), rnd((C,), 19, -0.3, 0.3)
    dz = rnd((rows, C), 20)

    # ---- reference ----
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    t = F.batch_norm(yr.t().reshape(1, C, rows), rm_ref, rv_ref, gr, br, True, 0.1, 1e-5).reshape(C, rows).t()
    if mode == "identity":
        t = t + r
    elif mode == "downsample":
        y2r = y2.clone().requires_grad_(True)
        g2r, b2r = g2.clone().requires_grad_(True), b2.clone().requires_grad_(True)
        t = t + F.batch_norm(y2r.t().reshape(1, C, rows), None, None, g2r, b2r, True, 0.1, 1e-5).reshape(C, rows).t()
    z_ref = torch.relu(t)
    z_ref.backward(dz)

    # ---- HIP: statistics come from a conv epilogue in production; emulate the partials layout [rows_blk][2][C] ----
    def coeffs(yy, gg, bb, rmean, rvar):
        blk = 128
        nb = (rows + blk - 1) // blk
        part = torch.zeros((nb, 2, C))
        for i in range(nb):
            sl = yy[i * blk:(i + 1) * blk]
            part[i, 0] = sl.sum(0)
            part[i, 1] = (sl * sl).sum(0)
        coef = torch.empty((4, C), device=DEV)
        wsb = hip.r3m_bn_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        rmd = None if rmean is None else rmean.to(DEV)
        rvd = None if rvar is None else rvar.to(DEV)
        partd, ggd, bbd = part.to(DEV), gg.to(DEV), bb.to(DEV)   # keep alive across the async launches
        rc = hip.r3m_bn_train_coeffs(partd.data_ptr(), nb, rows, ggd.data_ptr(), bbd.data_ptr(),
                                     None if rmd is None else rmd.data_ptr(), None if rvd is None else rvd.data_ptr(), 0.1, 1e-5,
                                     coef.data_ptr(), ws.data_ptr(), wsb, C, st())
        assert rc == 0, hip.r3m_last_error()
        torch.cuda.synchronize()
        return coef, rmd, rvd

    coef, rmd, rvd = coeffs(y, gamma, beta, rm, rv)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm_ref.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv_ref.numpy(), rtol=1e-5, atol=1e-6)
    yd, zd = y.to(DEV), torch.empty((rows, C), device=DEV)
    # 1-bit ReLU mask of the output (what the engine's backward reads instead of z) when rows*C is a multiple of 32
    bits = torch.zeros((rows * C + 31) // 32, dtype=torch.int32, device=DEV) if (rows * C) % 32 == 0 else None
    bits_ptr = None if bits is None else bits.data_ptr()
    coef2 = None
    if mode == "plain":
        rc = hip.r3m_bn_act_fwd(yd.data_ptr(), coef.data_ptr(), None, None, None, zd.data_ptr(), rows, C, 1, None, st())
    elif mode == "identity":
        rd = r.to(DEV)
        rc = hip.r3m_bn_act_fwd(yd.data_ptr(), coef.data_ptr(), rd.data_ptr(), None, None, zd.data_ptr(), rows, C, 1, bits_ptr, st())
    else:
        coef2, _, _ = coeffs(y2, g2, b2, None, None)
        y2d = y2.to(DEV)
        rc = hip.r3m_bn_act_fwd(yd.data_ptr(), coef.data_ptr(), None, y2d.data_ptr(), coef2.data_ptr(), zd.data_ptr(), rows, C, 1, bits_ptr, st())
    assert rc == 0, hip.r3m_last_error()
    e_max, _ = rel_err(zd.cpu().numpy(), z_ref.detach().numpy())
    assert e_max < 1e-5, f"bn fwd {e_max}"

    # ---- backward ----
    dzd = dz.to(DEV)
    wsb = hip.r3m_bn_workspace_bytes(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dg, db, dyd = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty((rows, C), device=DEV)
    zmask = None if mode == "plain" else zd
    use_bits = bits is not None and mode != "plain"
    if use_bits:   # the mask bits agree with z > 0 bit for bit
        zb = (zd.reshape(-1, 8, 4) > 0).to(torch.int64)
        w = (zb * (1 << (torch.arange(8, device=DEV).view(1, 8, 1) * 4 + torch.arange(4, device=DEV).view(1, 1, 4)))).sum((1, 2))
        assert torch.equal(w.to(torch.int32), bits)
    rc = hip.r3m_bn_bwd(dzd.data_ptr(), None if (zmask is None or use_bits) else zmask.data_ptr(), bits_ptr if use_bits else None,
                        yd.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), dyd.data_ptr(), ws.data_ptr(), wsb, rows, C, 1, 0,
                        st())
    assert rc == 0, hip.r3m_last_error()
    assert rel_err(dyd.cpu().numpy(), yr.grad.numpy())[0] < 2e-4
    assert rel_err(dg.cpu().numpy(), gr.grad.numpy())[0] < 1e-4
    assert rel_err(db.cpu().numpy(), br.grad.numpy())[0] < 1e-4
    if mode == "downsample":
        rc = hip.r3m_bn_bwd(dzd.data_ptr(), zd.data_ptr(), None, y2d.data_ptr(), coef2.data_ptr(), dg.data_ptr(), db.data_ptr(),
                            dyd.data_ptr(), ws.data_ptr(), wsb, rows, C, 1, 0, st())
        assert rc == 0
        assert rel_err(dyd.cpu().numpy(), y2r.grad.numpy())[0] < 2e-4
        assert rel_err(dg.cpu().numpy(), g2r.grad.numpy())[0] < 1e-4


def test_bn_eval(hip):
    rows, C = 500, 256
    y = rnd((rows, C), 31, -2, 2)
    gamma, beta, rm, rv = rnd((C,), 32, 0.5, 1.5), rnd((C,), 33, -0.3, 0.3), rnd((C,), 34, -0.2, 0.2), rnd((C,), 35, 0.5, 1.5)
    z_ref = torch.relu(F.batch_norm(y.t().reshape(1, C, rows), rm, rv, gamma, beta, False, 0.1, 1e-5).reshape(C, rows).t())
    coef = torch.empty((4, C), device=DEV)
    gd, bd, rmd, rvd, yd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), y.to(DEV)
    assert hip.r3m_bn_eval_coeffs(gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), 1e-5, coef.data_ptr(), C, st()) == 0
    zd = torch.empty((rows, C), device=DEV)
    assert hip.r3m_bn_act_fwd(yd.data_ptr(), coef.data_ptr(), None, None, None, zd.data_ptr(), rows, C, 1, None, st()) == 0
    assert rel_err(zd.cpu().numpy(), z_ref.numpy())[0] < 1e-5


@pytest.mark.parametrize("N,H,C", [(2, 112, 64), (3, 9, 64), (1, 10, 128)])
def test_maxpool(hip, N, H, C):
    z = torch.relu(rnd((N, C, H, H), 41))   # post-ReLU: plenty of exact-zero ties
    zr = z.clone().requires_grad_(True)
    p_ref = F.max_pool2d(zr, 3, 2, 1)
    Ho = p_ref.shape[2]
    dp = rnd(tuple(p_ref.shape), 42)
    p_ref.backward(dp)
    zd = nhwc(z).to(DEV)
    pd = torch.empty((N, Ho, Ho, C), device=DEV)
    am = torch.empty((N, Ho, Ho, C), dtype=torch.uint8, device=DEV)
    assert hip.r3m_maxpool_fwd(zd.data_ptr(), pd.data_ptr(), am.data_ptr(), N, H, H, C, st()) == 0
    torch.testing.assert_close(nchw(pd.cpu()), p_ref.detach(), rtol=0, atol=0)
    dzd = torch.empty((N, H, H, C), device=DEV)
    dpd = nhwc(dp).to(DEV)
    assert hip.r3m_maxpool_bwd(dpd.data_ptr(), am.data_ptr(), dzd.data_ptr(), N, H, H, C, st()) == 0
    # gradients may legitimately land on a different element of an all-zero (tied) window; after the ReLU mask they agree
    mask = (z > 0).float()
    torch.testing.assert_close(nchw(dzd.cpu()) * mask, zr.grad * mask, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(nchw(dzd.cpu()).sum(), zr.grad.sum(), rtol=1e-4, atol=1e-3)


def test_avgpool(hip):
    N, HW, C = 6, 49, 2048
    x = rnd((N, HW, C), 51)
    hd = torch.empty((N, C), device=DEV)
    xd = x.to(DEV)
    assert hip.r3m_avgpool_fwd(xd.data_ptr(), hd.data_ptr(), N, HW, C, st()) == 0
    ref = F.adaptive_avg_pool2d(x.permute(0, 2, 1).reshape(N, C, 7, 7), 1).flatten(1)
    assert rel_err(hd.cpu().numpy(), ref.numpy())[0] < 1e-6
    dh = rnd((N, C), 52)
    dxd = torch.empty((N, HW, C), device=DEV)
    dhd = dh.to(DEV)
    assert hip.r3m_avgpool_bwd(dhd.data_ptr(), dxd.data_ptr(), N, HW, C, st()) == 0
    torch.testing.assert_close(dxd.cpu(), (dh / 49.0).unsqueeze(1).expand(N, HW, C), rtol=1e-6, atol=1e-7)


def test_linear(hip):
    M, K, Nn = 120, 1792, 1024
    x, w, b = rnd((M, K), 61), rnd((Nn, K), 62, -0.05, 0.05), rnd((Nn,), 63)
    yd = torch.empty((M, Nn), device=DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    assert hip.r3m_linear_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), M, K, Nn, 1, st()) == 0
    ref = torch.relu(F.linear(x, w, b))
    assert rel_err(yd.cpu().numpy(), ref.numpy())[0] < 2e-5


def test_adam_matches_torch(hip):
    """G6: 3 steps vs torch.optim.Adam defaults (models_r3m.py:76)."""
    n = 4096 + 64
    p0 = rnd((n,), 71)
    grads = [rnd((n,), 72 + i, -0.01, 0.01) * (10.0 ** (i - 1)) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4)
    pd = p0.to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(grads):
        pr.grad = g.clone()
        opt.step()
        gd = g.to(DEV)
        assert hip.r3m_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, i + 1, 1.0, st()) == 0
        torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(m.cpu(), opt.state[pr]["exp_avg"], rtol=1e-6, atol=1e-12)
    torch.testing.assert_close(v.cpu(), opt.state[pr]["exp_avg_sq"], rtol=1e-6, atol=1e-14)


@pytest.mark.parametrize("dt", [0, 1], ids=["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,C", [(2, 112, 112, 64), (3, 9, 9, 64), (1, 10, 10, 128), (2, 12, 9, 64), (2, 7, 16, 32)])
def test_stem_tail_fused_equals_unfused(hip, N, H, W, C, dt):
    """BN+ReLU+MaxPool fused (what the engine runs) vs the separate operators, bit for bit: pooled values, argmax bytes,
    dY, dgamma, dbeta. Post-BN values have plenty of exact zeros (ReLU) -> ties in the windows are exercised."""
    tdt = torch.float32 if dt == 0 else torch.bfloat16
    rows = N * H * W
    y = rnd((rows, C), 61, -2.0, 2.0).to(DEV).to(tdt)
    yf = y.float()
    m, v = yf.mean(0), yf.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(v + 1e-5)
    gam, bet = rnd((C,), 62, 0.5, 1.5).to(DEV), rnd((C,), 63, -0.3, 0.3).to(DEV)
    coef = torch.stack([m, inv, gam * inv, bet - m * gam * inv]).contiguous()
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    # unfused
    z = torch.empty((rows, C), dtype=tdt, device=DEV)
    assert hip.r3m_bn_act_fwd_dt(y.data_ptr(), coef.data_ptr(), None, None, None, z.data_ptr(), rows, C, 1, None, dt, st()) == 0
    p0 = torch.empty((N, Ho, Wo, C), dtype=tdt, device=DEV)
    a0 = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=DEV)
    assert hip.r3m_maxpool_fwd_dt(z.data_ptr(), p0.data_ptr(), a0.data_ptr(), N, H, W, C, dt, st()) == 0
    # fused
    p1 = torch.empty_like(p0)
    a1 = torch.empty_like(a0)
    assert hip.r3m_bn_relu_maxpool_fwd_dt(y.data_ptr(), coef.data_ptr(), p1.data_ptr(), a1.data_ptr(), N, H, W, C, dt, st()) == 0, hip.r3m_last_error()
    assert torch.equal(p0, p1) and torch.equal(a0, a1)
    # backward
    dp = rnd((N, Ho, Wo, C), 64).to(DEV).to(tdt)
    dz = torch.empty((rows, C), dtype=tdt, device=DEV)
    assert hip.r3m_maxpool_bwd_dt(dp.data_ptr(), a0.data_ptr(), dz.data_ptr(), N, H, W, C, dt, st()) == 0
    wsb = hip.r3m_bn_workspace_bytes(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dg0, db0, dy0 = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty((rows, C), dtype=tdt, device=DEV)
    assert hip.r3m_bn_bwd_dt(dz.data_ptr(), None, None, y.data_ptr(), coef.data_ptr(), dg0.data_ptr(), db0.data_ptr(), dy0.data_ptr(),
                             ws.data_ptr(), wsb, rows, C, 1, 0, dt, st()) == 0
    dg1, db1, dy1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty((rows, C), dtype=tdt, device=DEV)
    assert hip.r3m_bn_maxpool_bwd_dt(dp.data_ptr(), a1.data_ptr(), y.data_ptr(), coef.data_ptr(), dg1.data_ptr(), db1.data_ptr(),
                                     dy1.data_ptr(), ws.data_ptr(), wsb, N, H, W, C, 1, 0, dt, st()) == 0, hip.r3m_last_error()
    # the parameter gradients are sums in a different (but fixed) order between the two reduce kernels
    assert rel_err(dg1.cpu().numpy(), dg0.cpu().numpy())[0] < 1e-5 and rel_err(db1.cpu().numpy(), db0.cpu().numpy())[0] < 1e-5
    assert rel_err(dy1.float().cpu().numpy(), dy0.float().cpu().numpy())[0] < (1e-5 if dt == 0 else 2.0 ** -7)


@pytest.mark.parametrize("cfg", [dict(momentum=0.0), dict(momentum=0.9), dict(momentum=0.9, dampening=0.1, weight_decay=1e-2),
                                 dict(momentum=0.9, nesterov=True, weight_decay=1e-3)], ids=["plain", "momentum", "damp_wd", "nesterov"])
def test_sgd_matches_torch(hip, cfg):
    """the plain optimizer of north_star's 'SGD/Adam step': 3 steps vs torch.optim.SGD on the same flat buffer"""
    n = 4096 + 64
    p0 = rnd((n,), 81)
    grads = [rnd((n,), 82 + i, -0.01, 0.01) * (10.0 ** (i - 1)) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=1e-2, **cfg)
    pd = p0.to(DEV)
    buf = torch.full((n,), float("nan"), device=DEV)          # must be initialised by step 1, not read
    for i, g in enumerate(grads):
        pr.grad = g.clone()
        opt.step()
        gd = g.to(DEV)
        rc = hip.r3m_sgd_step(pd.data_ptr(), gd.data_ptr(), buf.data_ptr() if cfg.get("momentum", 0.0) else None, n, 1e-2,
                              cfg.get("momentum", 0.0), cfg.get("dampening", 0.0), cfg.get("weight_decay", 0.0),
                              1 if cfg.get("nesterov") else 0, i + 1, 1.0, st())
        assert rc == 0, hip.r3m_last_error()
        torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-6, atol=1e-9)
    if cfg.get("momentum", 0.0):
        ref_buf = opt.state[pr]["momentum_buffer"]     # fp32 round-off level of the buffer's scale (fma contraction differs)
        torch.testing.assert_close(buf.cpu(), ref_buf, rtol=1e-6, atol=2e-7 * float(ref_buf.abs().max()))
    assert hip.r3m_sgd_step(pd.data_ptr(), gd.data_ptr(), None, n, 1e-2, 0.9, 0.0, 0.0, 0, 1, 1.0, st()) != 0   # momentum without a buffer


def test_adam_matches_committed_golden(hip, golden_dir):
    """G6: three fused-Adam steps against the committed torch.optim.Adam trajectory (tests/golden/adam.npz)."""
    import os
    from oracle import detgen
    g = np.load(os.path.join(golden_dir, "adam.npz"))
    n = 4096
    p = torch.from_numpy(detgen.uniform("g6p", (n,), -1.0, 1.0)).to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(3):
        gr = torch.from_numpy(detgen.uniform(f"g6g{i}", (n,), -1.0, 1.0) * np.float32(10.0 ** (i - 1))).to(DEV)
        assert hip.r3m_adam_step(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, i + 1, 1.0, st()) == 0
        np.testing.assert_allclose(p.cpu().numpy(), g[f"p_{i}"], rtol=1e-6, atol=2e-7)
    np.testing.assert_allclose(m.cpu().numpy(), g["exp_avg"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), g["exp_avg_sq"], rtol=1e-6, atol=1e-12)


def _pack_bits(mask_nhwc):
    """[..] bool (NHWC element order) -> uint32 words, bit i of word w = element 32 w + i (the engine's 1-bit ReLU mask layout)."""
    flat = mask_nhwc.reshape(-1).to(torch.int64)
    pad = (-flat.numel()) % 32
    if pad:
        flat = torch.cat([flat, torch.zeros(pad, dtype=torch.int64)])
    words = (flat.view(-1, 32) << torch.arange(32, dtype=torch.int64)).sum(1)
    return (words & 0xFFFFFFFF).to(torch.int64).numpy().astype(np.uint32)


BNRED_CASES = [(2, 56, 64, 256, 1, 1, 0), (2, 56, 256, 64, 1, 1, 0), (2, 28, 128, 128, 3, 1, 1), (2, 56, 128, 128, 3, 2, 1),
               (3, 14, 256, 256, 3, 1, 1), (5, 7, 2048, 512, 1, 1, 0), (2, 56, 64, 128, 3, 2, 1), (3, 9, 64, 64, 3, 1, 1),
               (7, 5, 160, 64, 1, 1, 0), (2, 13, 32, 64, 3, 2, 1),
               # persistent 1x1 kernel: several tiles per worker, partial last row tile (dgrad: GEMM N = Ci, K = Co)
               (24, 56, 256, 64, 1, 1, 0), (21, 56, 256, 128, 1, 1, 0), (24, 56, 64, 256, 1, 1, 0), (23, 56, 64, 128, 1, 1, 0),
               # ... strided-output form: a stride-2 dgrad's parity classes, more tiles than workers
               (90, 56, 128, 128, 3, 2, 1), (91, 56, 64, 64, 3, 2, 1)]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["recompute", "bits", "bits+residual"])
@pytest.mark.parametrize("case", BNRED_CASES, ids=_ids)
def test_dgrad_epilogue_emits_bn_backward_partials(hip, case, mode, dtype):
    """EPI_BNRED (round 2): the dgrad epilogue also accumulates the consumer BatchNorm's backward partials — per 64 result rows
    sum(g) and sum(g (y - mean)), g = dz * [ReLU mask] — so the stand-alone reduce pass (one more read of dz and y) disappears
    from the engine. Checked against float64 on the STORED dz (one rounding for bf16) for the three forms the engine uses:
    mask recomputed from y (inner BatchNorms), mask bits (block outputs), mask bits + masked residual-gradient join; stride 1 and
    2 (four parity launches appending their partial rows), partial tiles, Ci = 32..2048."""
    N, H, Ci, Co, k, s, p = case
    if dtype == "bf16" and (Ci % 64 or Co % 64):
        pytest.skip("bf16 kernels: channel counts are multiples of 64")
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    dt = 0 if dtype == "fp32" else 1
    q = (lambda t: t) if dtype == "fp32" else (lambda t: t.to(torch.bfloat16).float())
    Ho = (H + 2 * p - k) // s + 1
    w = q(rnd((Co, Ci, k, k), 2, -0.2, 0.2))
    dy = q(rnd((N, Co, Ho, Ho), 3))
    y = q(rnd((N, Ci, H, H), 4, -1.0, 1.5))                                  # the consumer BatchNorm's input
    res = q(rnd((N, Ci, H, H), 5))                                           # residual gradient joining at this tensor
    res_mask = rnd((N, Ci, H, H), 6) > 0.0
    scale = rnd((Ci,), 7, 0.5, 1.5)
    shift = rnd((Ci,), 8, -0.5, 0.5)
    mean = rnd((Ci,), 9, -0.2, 0.4)
    bn_mask_bits = rnd((N, Ci, H, H), 10) > -0.3
    # recomputed mask = fmaf(y, scale, shift) > 0 in fp32: keep every element away from the kink so the decision is unambiguous
    for _ in range(4):
        v = y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        y = q(torch.where(v.abs() < 1e-3, y + 0.05, y))
    assert not bool(((y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).abs() < 1e-5).any())
    # float64 expectation
    xr = torch.zeros((N, Ci, H, H), dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w.double(), stride=s, padding=p).backward(dy.double())
    dz = xr.grad
    if mode == "bits+residual":
        dz = dz + res.double() * res_mask
    dz_stored = dz.float() if dtype == "fp32" else dz.to(torch.bfloat16).float()
    on = bn_mask_bits if mode != "recompute" else (y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) > 0
    g = dz_stored.double() * on
    exp_s1 = g.sum((0, 2, 3)).numpy()
    exp_s2 = (g * (y.double() - mean.double().view(1, -1, 1, 1))).sum((0, 2, 3)).numpy()
    # device
    dyd, yd = nhwc(dy).to(DEV).to(tdt), nhwc(y).to(DEV).to(tdt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)                           # fp32 master (dgrad converts for bf16)
    dxd = torch.full((N, H, H, Ci), float("nan"), device=DEV, dtype=tdt)
    wsb = hip.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rows = hip.r3m_conv2d_dgrad_bnred_rows(N, H, H, s)
    part = torch.full((rows, 2, Ci), float("nan"), device=DEV)
    resd = nhwc(res).to(DEV).to(tdt) if mode == "bits+residual" else None
    resb = torch.from_numpy(_pack_bits(nhwc(res_mask)).astype(np.int64)).to(torch.int32).to(DEV) if mode == "bits+residual" else None
    bnb = None if mode == "recompute" else torch.from_numpy(_pack_bits(nhwc(bn_mask_bits)).astype(np.int64)).to(torch.int32).to(DEV)
    sc, sh, mu = scale.to(DEV), shift.to(DEV), mean.to(DEV)
    ptr = lambda t: None if t is None else t.data_ptr()
    rc = hip.r3m_conv2d_dgrad_bnred_dt(dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p,
                                       ptr(resd), ptr(resb), yd.data_ptr(), ptr(bnb), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                                       part.data_ptr(), dt, st())
    assert rc == 0, hip.r3m_last_error()
    dx = nchw(dxd.float().cpu())
    tol = 2e-5 if dtype == "fp32" else 2.0 ** -8
    e_dx = rel_err(dx.numpy(), dz.numpy())[0]
    if not e_dx < tol:      # say WHERE: whole tiles missing (scheduling), single rows (addressing) or everything (arithmetic)
        d = (nhwc(dx).double() - nhwc(dz.double())).abs().reshape(-1, Ci)
        bad = (~torch.isfinite(d)) | (d > tol * float(dz.abs().max()))
        rows = torch.nonzero(bad.any(1)).flatten()
        cols = torch.nonzero(bad.any(0)).flatten()
        pytest.fail(f"dgrad result: max-rel {e_dx}; non-finite {int((~torch.isfinite(d)).sum())}; {len(rows)} bad rows of {d.shape[0]} "
                    f"(first {rows[:6].tolist()}, last {rows[-3:].tolist()}, 128-row tiles {sorted(set((rows // 128).tolist()))[:12]}); "
                    f"{len(cols)} bad columns (first {cols[:6].tolist()}, last {cols[-3:].tolist()})")
    assert torch.isfinite(part).all(), "a partial row was not written"
    # the partials must describe the dz the kernel STORED (bf16: its own rounding of its own fp32 sum)
    g_dev = dx.double() * on
    got_s1 = part[:, 0].double().sum(0).cpu().numpy()
    got_s2 = part[:, 1].double().sum(0).cpu().numpy()
    ref_s1 = g_dev.sum((0, 2, 3)).numpy()
    ref_s2 = (g_dev * (y.double() - mean.double().view(1, -1, 1, 1))).sum((0, 2, 3)).numpy()
    scale1 = float(g_dev.abs().sum((0, 2, 3)).max())
    np.testing.assert_allclose(got_s1, ref_s1, rtol=0, atol=2e-6 * scale1)
    np.testing.assert_allclose(got_s2, ref_s2, rtol=0, atol=4e-6 * scale1)
    # and agree with the float64 expectation to the accuracy of the stored tensor
    np.testing.assert_allclose(got_s1, exp_s1, rtol=0, atol=(2e-5 if dtype == "fp32" else 2e-2) * scale1 / np.sqrt(N * H * H) + 1e-6 * scale1)
