#!/usr/bin/env python
"""GPU: the round-6 bf16 3x3 kernels (csrc/conv_row16.hip, r3m_debug_set_conv3x3_bf16(1)) against the per-tile halo kernels they
replace (mode 0) and against a float64 reference on the bf16-rounded operands, launch by launch through the C ABI.
    row16_check.py ops    forward (+ BatchNorm statistics), plain dgrad, dgrad with the EPI_BNRED (+ masked residual join) epilogue
    row16_check.py bench  per-launch times of both modes on the bench shapes (interleaved repetitions)
"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import _lib

L = _lib.lib()
P = lambda t: None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def run_fwd(x, w, N, H, Ci, Co):
    y = torch.empty((N, H, H, Co), device="cuda", dtype=torch.bfloat16)
    rows = L.r3m_conv2d_stats_rows(N, H, H, Co, 3, 1, 1)
    stats = torch.zeros((rows, 2, Co), device="cuda")
    assert L.r3m_conv2d_fwd_dt(P(x), P(w), P(y), P(stats), N, H, H, Ci, Co, 3, 1, 1, 1, stream()) == 0, L.r3m_last_error()
    return y, stats


def run_dgrad(dy, wf, N, H, Ci, Co, bn=None):
    dx = torch.empty((N, H, H, Ci), device="cuda", dtype=torch.bfloat16)
    wsb = L.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, 3)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    if bn is None:
        assert L.r3m_conv2d_dgrad_dt(P(dy), P(wf), P(dx), P(ws), wsb, N, H, H, Ci, Co, 3, 1, 1, 1, stream()) == 0, L.r3m_last_error()
        return dx, None
    res, rbits, by, ybits, sc, sh, mu = bn
    rows = L.r3m_conv2d_dgrad_bnred_rows(N, H, H, 1)
    part = torch.zeros((rows, 2, Ci), device="cuda")
    assert L.r3m_conv2d_dgrad_bnred_dt(P(dy), P(wf), P(dx), P(ws), wsb, N, H, H, Ci, Co, 3, 1, 1, P(res), P(rbits), P(by), P(ybits),
                                       P(sc), P(sh), P(mu), P(part), 1, stream()) == 0, L.r3m_last_error()
    return dx, part


def ulp_report(name, a, b, ref):
    """a, b: bf16 tensors of the two kernels; ref: float64 truth. Reports the share of elements that differ, the largest difference in
    bf16 ulps of the element, and each kernel's own error against the truth in ulps."""
    af, bf = a.double(), b.double()
    ulp = torch.maximum(ref.abs(), 0.25 * ref.pow(2).mean().sqrt())     # (a cancelled sum is not judged against its own tiny value)
    ulp = torch.exp2(torch.floor(torch.log2(ulp)) - 7)                        # bf16: 8 significant bits
    d = ((af - bf).abs() / ulp)
    ea, eb = ((af - ref).abs() / ulp).max().item(), ((bf - ref).abs() / ulp).max().item()
    ndiff = (a != b).float().mean().item()
    print(f"  {name}: differ {100 * ndiff:.3f} % of elements, max |new - halo| = {d.max().item():.2f} ulp; vs float64: new {ea:.2f} ulp, halo {eb:.2f} ulp")
    return d.max().item(), ea, eb


def ops():
    torch.manual_seed(0)
    worst = 0.0
    cases = [(5, 28, 128, 128), (3, 14, 256, 256), (9, 7, 512, 512), (40, 14, 128, 256), (2, 28, 64, 128), (130, 7, 256, 512), (1, 30, 128, 128),
             (2, 56, 64, 64), (3, 28, 128, 64), (1, 9, 64, 64), (5, 56, 64, 64)]
    for (N, H, Ci, Co) in cases:
        print(f"case N={N} H={H} Ci={Ci} Co={Co}  (M = {N * H * H})")
        x = torch.randn((N, H, H, Ci), device="cuda").bfloat16()
        w = (torch.randn((Co, 3, 3, Ci), device="cuda") * (1.5 / (9 * Ci) ** 0.5)).bfloat16()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).contiguous()
        out = {}
        for mode in (1, 0):
            L.r3m_debug_set_conv3x3_bf16(mode)
            out[mode] = run_fwd(x, w, N, H, Ci, Co)
        torch.cuda.synchronize()
        d, ea, eb = ulp_report("fwd y", out[1][0], out[0][0], ref)
        worst = max(worst, d)
        assert d <= 1.0 and ea <= 0.5 + 1e-3 + (d > 0) * 0.01, "forward: more than one rounding away"
        # statistics: fp32 column sums of the fp32 accumulators per 128 rows: compare with float64 sums of the truth
        M = N * H * H
        SR = 128 if Co % 128 == 0 else 256                 # rows per statistics partial row
        s_ref = torch.stack([ref.reshape(M, Co)[r:r + SR].sum(0) for r in range(0, M, SR)])
        q_ref = torch.stack([(ref.reshape(M, Co)[r:r + SR] ** 2).sum(0) for r in range(0, M, SR)])
        for mode in (1, 0):
            st = out[mode][1].double()
            assert st.shape[0] == s_ref.shape[0], (st.shape, s_ref.shape)
            e1 = ((st[:, 0] - s_ref).abs().max() / s_ref.abs().max()).item()
            e2 = ((st[:, 1] - q_ref).abs().max() / q_ref.abs().max()).item()
            print(f"  stats mode {mode}: rel err sum {e1:.2e}, sum of squares {e2:.2e}")
            assert e1 < 2e-6 and e2 < 2e-6
        # dgrad: dy [N,H,H,Co] -> dx [N,H,H,Ci]
        dy = torch.randn((N, H, H, Co), device="cuda").bfloat16()
        wf = w.float()
        dref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).contiguous()
        dg = {}
        for mode in (1, 0):
            L.r3m_debug_set_conv3x3_bf16(mode)
            dg[mode] = run_dgrad(dy, wf, N, H, Ci, Co)[0]
        torch.cuda.synchronize()
        d, ea, eb = ulp_report("dgrad dx", dg[1], dg[0], dref)
        worst = max(worst, d)
        assert d <= 1.0 and ea <= 0.51
        # dgrad + EPI_BNRED (+ masked residual join with mask bits)
        nbits = N * H * H * Ci // 32
        for joined in (False, True):
            by = torch.randn((N, H, H, Ci), device="cuda").bfloat16()
            res = torch.randn((N, H, H, Ci), device="cuda").bfloat16() if joined else None
            rbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (nbits,), device="cuda", dtype=torch.int32) if joined else None
            ybits = torch.randint(-2 ** 31, 2 ** 31 - 1, (nbits,), device="cuda", dtype=torch.int32) if joined else None
            sc, sh, mu = torch.rand(Ci, device="cuda") + 0.5, torch.rand(Ci, device="cuda") - 0.5, torch.rand(Ci, device="cuda")
            r = {}
            for mode in (1, 0):
                L.r3m_debug_set_conv3x3_bf16(mode)
                r[mode] = run_dgrad(dy, wf, N, H, Ci, Co, (res, rbits, by, ybits, sc, sh, mu))
            torch.cuda.synchronize()
            dd = ((r[1][0].double() - r[0][0].double()).abs() / (r[0][0].double().abs() + 1e-3)).max().item()
            pe = ((r[1][1] - r[0][1]).abs().max() / r[0][1].abs().max()).item()
            print(f"  dgrad+bnred{'+join' if joined else ''}: max rel |new - halo| dx {dd:.2e} (one bf16 ulp = 7.8e-3), partials {pe:.2e}")
            assert dd <= 8e-3 and pe < 2e-3
    L.r3m_debug_set_conv3x3_bf16(1)
    print(f"OK (largest new-vs-halo difference {worst:.2f} bf16 ulp)")


def bench():
    shapes = [(2560, 56, 64, 64), (2560, 28, 128, 128), (2560, 14, 256, 256), (2560, 7, 512, 512), (1280, 56, 64, 64), (1280, 28, 128, 128), (1280, 14, 256, 256), (1280, 7, 512, 512)]
    reps = int(os.environ.get("REPS", "3"))
    data = os.environ.get("DATA", "randn")       # randn | zeros | ones: the same instruction stream on low-toggle operands (DVFS check)
    for (N, H, Ci, Co) in shapes:
        x = torch.randn((N, H, H, Ci), device="cuda").bfloat16()
        w = (torch.randn((Co, 3, 3, Ci), device="cuda") * 0.05).bfloat16()
        if data == "zeros":
            x.zero_(); w.zero_()
        elif data == "ones":
            x.fill_(1.0); w.fill_(0.01)
        y = torch.empty((N, H, H, Co), device="cuda", dtype=torch.bfloat16)
        rows = L.r3m_conv2d_stats_rows(N, H, H, Co, 3, 1, 1)
        stats = torch.zeros((rows, 2, Co), device="cuda")
        flops = 2.0 * N * H * H * Co * Ci * 9
        fn = lambda: L.r3m_conv2d_fwd_dt(P(x), P(w), P(y), P(stats), N, H, H, Ci, Co, 3, 1, 1, 1, stream())
        line = f"fwd16+stats[{data}] {N:5d}x{H:2d}x{H:2d} {Ci:3d}->{Co:3d}: "
        for rep in range(reps):
            for mode in ((1,) if os.environ.get("ONLY_NEW") else (0, 1)):
                L.r3m_debug_set_conv3x3_bf16(mode)
                for _ in range(3):
                    assert fn() == 0, L.r3m_last_error()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                line += f" {'halo' if mode == 0 else 'row '} {ms:.3f} ms ({flops / ms / 1e9:6.0f} TF/s)"
        print(line)
    L.r3m_debug_set_conv3x3_bf16(1)


if __name__ == "__main__":
    {"ops": ops, "bench": bench}[sys.argv[1]]()
