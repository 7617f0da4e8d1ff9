import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from r3m_amd import _lib
L = _lib.lib()
torch.manual_seed(1)
for (N, H, Ci, Co) in [(2, 8, 64, 64), (3, 9, 64, 64), (2, 8, 128, 128), (4, 14, 128, 128)]:
    x = torch.randn(N, H, H, Ci); dy = torch.randn(N, H, H, Co)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).backward(dy.permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).contiguous()          # [Co][3][3][Ci]
    xd, dyd = x.cuda(), dy.cuda()
    dw = torch.full((Co, 3, 3, Ci), float("nan"), device="cuda")
    wsb = L.r3m_conv2d_wgrad_workspace_bytes_dt(N, H, H, Ci, Co, 3, 1, 1, 0)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.r3m_conv2d_wgrad_dt(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, 3, 1, 1, 0, 0, st) == 0
    got = dw.cpu()
    print("case", N, H, Ci, Co, "max ref", ref.abs().max().item())
    for kh in range(3):
        print("  kh", kh, ["%.3g" % (got[:, kh, kw] - ref[:, kh, kw]).abs().max().item() for kw in range(3)],
              "got absmax", ["%.3g" % got[:, kh, kw].abs().max().item() for kw in range(3)])
    # does a tap match another tap of the reference?
    for kh in range(3):
        for kw in range(3):
            best = min(((got[:, kh, kw] - ref[:, a, b]).abs().max().item(), a, b) for a in range(3) for b in range(3))
            print("    got[%d,%d] closest ref tap (%d,%d) err %.3g" % (kh, kw, best[1], best[2], best[0]))
    # channel structure of the error for tap (1,1)
    e = (got[:, 1, 1] - ref[:, 1, 1]).abs()
    print("  tap(1,1) bad co rows", (e.max(1)[0] > 1e-3).nonzero().flatten()[:16].tolist(), "bad ci cols", (e.max(0)[0] > 1e-3).nonzero().flatten()[:16].tolist())
