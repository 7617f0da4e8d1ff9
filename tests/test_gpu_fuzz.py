"""-m gpu: seeded random convolution geometries through the C ABI against `F.conv2d` autograd on the CPU (the arithmetic under
torchvision's ResNet.forward, call site /root/reference/r3m/models/models_r3m.py:99). Round 3 rewrote the operand addressing of
every fp32 weight-gradient launch (buffer descriptors, scalar row cursor, kernel-row blocks) and added an input-window kernel for
3x3 / stride-1 launches: their border, tail and out-of-range logic is what odd sizes exercise — image widths 3..29 (window rows
wrap inside a tile, tiles span several frames), frame counts that leave partial tiles and partial K steps, 32 / 64 / 96 / 128 /
160 / 256-channel sides (narrow, wide and odd-chunk routes), strides 1 and 2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import nchw, nhwc, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def st():
    return torch.cuda.current_stream().cuda_stream


def _cases(n, seed):
    rng = np.random.RandomState(seed)
    chans = [32, 64, 96, 128, 160, 256]
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 3, 3]))
        s = int(rng.choice([1, 1, 2]))
        H = int(rng.randint(3, 30))
        N = int(rng.randint(1, 7))
        Ci, Co = int(rng.choice(chans)), int(rng.choice(chans))
        if (H + 2 * (k // 2) - k) // s + 1 < 1:
            continue
        out.append((N, H, Ci, Co, k, s, k // 2))
    return out


@pytest.mark.parametrize("case", _cases(36, 20260928), ids=lambda c: "N{}_H{}_{}to{}_k{}s{}p{}".format(*c))
def test_random_conv_geometry_fwd_dgrad_wgrad(hip, case):
    N, H, Ci, Co, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.rand((N, Ci, H, H), generator=g) * 2 - 1
    w = (torch.rand((Co, Ci, k, k), generator=g) * 2 - 1) * 0.2
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=p)
    Ho = y_ref.shape[2]
    dy = torch.rand(tuple(y_ref.shape), generator=g) * 2 - 1
    y_ref.backward(dy)

    xd, wd = nhwc(x).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    yd = torch.full((N, Ho, Ho, Co), float("nan"), device=DEV)
    rows = hip.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
    stats = torch.zeros((rows, 2, Co), device=DEV)
    assert hip.r3m_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, st()) == 0, hip.r3m_last_error()
    assert rel_err(nchw(yd.cpu()).numpy(), y_ref.detach().numpy())[0] < 2e-5
    yr = y_ref.detach().double()
    np.testing.assert_allclose(stats[:, 0].double().sum(0).cpu().numpy(), yr.sum((0, 2, 3)).numpy(), rtol=1e-4,
                               atol=1e-3 * float(yr.abs().max()) * max(1.0, N * Ho * Ho / 64.0))

    dyd = nhwc(dy).to(DEV)
    dxd = torch.full((N, H, H, Ci), float("nan"), device=DEV)
    wsb = hip.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    assert hip.r3m_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, st()) == 0, hip.r3m_last_error()
    assert rel_err(nchw(dxd.cpu()).numpy(), xr.grad.numpy())[0] < 2e-5

    dwd = torch.full((Co, k, k, Ci), float("nan"), device=DEV)
    wsb = hip.r3m_conv2d_wgrad_workspace_bytes(N, H, H, Ci, Co, k, s, p)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    assert hip.r3m_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 0, st()) == 0, hip.r3m_last_error()
    assert rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), wr.grad.numpy())[0] < 5e-5


@pytest.mark.parametrize("case", [c for c in _cases(60, 7) if c[2] % 64 == 0 and c[3] % 64 == 0][:14],
                         ids=lambda c: "N{}_H{}_{}to{}_k{}s{}p{}".format(*c))
def test_random_conv_geometry_bf16(hip, case):
    """The bf16 entry points on the same kind of geometry: against float64 on the bf16-rounded operands, one rounding of the output
    range for bf16 results, fp32 level for the weight gradient (fp32 accumulation of exact bf16 products)."""
    N, H, Ci, Co, k, s, p = case
    BF16 = 1
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = (torch.rand((N, Ci, H, H), generator=g) * 2 - 1).bfloat16()
    w = ((torch.rand((Co, Ci, k, k), generator=g) * 2 - 1) * 0.2)
    wb = w.bfloat16()
    xr, wr = x.double().requires_grad_(True), wb.double().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=p)
    Ho = y_ref.shape[2]
    dy = (torch.rand(tuple(y_ref.shape), generator=g) * 2 - 1).bfloat16()
    y_ref.backward(dy.double())

    xd = nhwc(x.float()).bfloat16().to(DEV)
    wd = wb.permute(0, 2, 3, 1).contiguous().to(DEV)
    yd = torch.zeros((N, Ho, Ho, Co), dtype=torch.bfloat16, device=DEV)
    assert hip.r3m_conv2d_fwd_dt(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), None, N, H, H, Ci, Co, k, s, p, BF16, st()) == 0, hip.r3m_last_error()
    assert rel_err(nchw(yd.float().cpu()).numpy(), y_ref.detach().numpy())[0] < 2.0 ** -7

    dyd = nhwc(dy.float()).bfloat16().to(DEV)
    dwd = torch.full((Co, k, k, Ci), float("nan"), device=DEV)
    wsb = hip.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, k, s, p, BF16)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    assert hip.r3m_conv2d_wgrad_dt(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 0, BF16, st()) == 0, hip.r3m_last_error()
    assert rel_err(dwd.cpu().permute(0, 3, 1, 2).numpy(), wr.grad.numpy())[0] < 5e-5
