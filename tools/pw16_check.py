#!/usr/bin/env python
"""GPU: the persistent warp-specialised bf16 kernel (csrc/conv_pw16.hip) against the per-tile kernels it replaces, launch by launch.

    pw16_check.py ops    every conv case below, forward (+ BatchNorm statistics) and input gradient, through the C ABI with
                         r3m_debug_set_pw16(0) and (1): results must be BIT-IDENTICAL (same MFMA order, same fp32 accumulators, one
                         rounding), then both are timed
    pw16_check.py step   one training step of BASELINE configs[2] / configs[4] with the switch off and on, interleaved repetitions
"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3m_amd import _lib

L = _lib.lib()
DEV = "cuda:0"
MODE = int(os.environ.get("PW16_MODE", "1"))     # r3m_debug_set_pw16: 1 = pointwise / gather forms, 3 = + the 3x3 window form


def st():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (N, H, Ci, Co, k, stride, pad): ResNet-50 / 34 geometries at test sizes + at the bench sizes (1280 / 2560 frames)
SMALL = [(2, 56, 64, 64, 1, 1, 0), (2, 56, 64, 256, 1, 1, 0), (2, 56, 256, 64, 1, 1, 0), (2, 56, 256, 128, 1, 1, 0),
         (2, 56, 128, 128, 3, 2, 1), (2, 28, 128, 512, 1, 1, 0), (2, 56, 256, 512, 1, 2, 0), (2, 28, 512, 128, 1, 1, 0),
         (3, 14, 256, 1024, 1, 1, 0), (2, 28, 512, 1024, 1, 2, 0), (3, 14, 1024, 256, 1, 1, 0), (3, 14, 512, 512, 3, 2, 1),
         (5, 7, 512, 2048, 1, 1, 0), (3, 14, 1024, 2048, 1, 2, 0), (5, 7, 2048, 512, 1, 1, 0), (2, 56, 64, 128, 3, 2, 1),
         (2, 56, 64, 128, 1, 2, 0), (1, 11, 64, 192, 3, 2, 1), (7, 5, 192, 64, 1, 1, 0), (40, 56, 64, 256, 1, 1, 0),
         (40, 56, 256, 64, 1, 1, 0), (33, 28, 128, 512, 1, 1, 0), (40, 28, 256, 256, 3, 2, 1),
         (2, 56, 64, 64, 3, 1, 1), (2, 28, 128, 128, 3, 1, 1), (3, 14, 256, 256, 3, 1, 1), (5, 7, 512, 512, 3, 1, 1), (3, 9, 64, 64, 3, 1, 1),
         (84, 28, 128, 128, 3, 1, 1), (3, 8, 64, 128, 3, 1, 1), (5, 16, 128, 64, 3, 1, 1), (40, 14, 256, 256, 3, 1, 1), (1, 5, 64, 256, 3, 1, 1),
         (37, 7, 512, 512, 3, 1, 1), (9, 56, 64, 64, 3, 1, 1)]
BIG = [(1280, 56, 64, 256, 1, 1, 0), (1280, 56, 256, 64, 1, 1, 0), (1280, 56, 64, 64, 1, 1, 0), (1280, 28, 128, 512, 1, 1, 0),
       (1280, 28, 512, 128, 1, 1, 0), (1280, 14, 256, 1024, 1, 1, 0), (1280, 14, 1024, 256, 1, 1, 0), (1280, 7, 512, 2048, 1, 1, 0),
       (1280, 7, 2048, 512, 1, 1, 0), (1280, 56, 256, 128, 1, 1, 0), (1280, 56, 128, 128, 3, 2, 1), (1280, 56, 256, 512, 1, 2, 0),
       (1280, 28, 256, 256, 3, 2, 1), (1280, 28, 512, 1024, 1, 2, 0), (1280, 14, 512, 512, 3, 2, 1), (1280, 14, 1024, 2048, 1, 2, 0),
       (1280, 56, 64, 64, 3, 1, 1), (1280, 28, 128, 128, 3, 1, 1), (1280, 14, 256, 256, 3, 1, 1), (1280, 7, 512, 512, 3, 1, 1),
       (2560, 56, 64, 64, 3, 1, 1), (2560, 28, 128, 128, 3, 1, 1), (2560, 14, 256, 256, 3, 1, 1), (2560, 7, 512, 512, 3, 1, 1)]


def run_case(case, do_time):
    N, H, Ci, Co, k, s, p = case
    Ho = (H + 2 * p - k) // s + 1
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn((N, H, H, Ci), device=DEV, generator=g).bfloat16()
    w32 = torch.randn((Co, k, k, Ci), device=DEV, generator=g) * 0.05
    w = w32.bfloat16()
    dy = torch.randn((N, Ho, Ho, Co), device=DEV, generator=g).bfloat16()
    res = torch.randn((N, H, H, Ci), device=DEV, generator=g).bfloat16()
    rbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (x.numel() // 32 + 1,), device=DEV, dtype=torch.int32, generator=g)
    rows = L.r3m_conv2d_stats_rows(N, H, H, Co, k, s, p)
    wsb = L.r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    outs, times = {}, {}
    for mode in (0, 1):
        L.r3m_debug_set_pw16(MODE if mode else 0)
        y = torch.full((N, Ho, Ho, Co), float("nan"), device=DEV).bfloat16()
        stats = torch.full((rows, 2, Co), float("nan"), device=DEV)
        dx = torch.full((N, H, H, Ci), float("nan"), device=DEV).bfloat16()
        dxr = torch.full((N, H, H, Ci), float("nan"), device=DEV).bfloat16()
        fwd = lambda: L.r3m_conv2d_fwd_dt(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, H, Ci, Co, k, s, p, 1, st())
        dg = lambda: L.r3m_conv2d_dgrad_dt(dy.data_ptr(), w32.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsb, N, H, H, Ci, Co, k, s, p, 1, st())
        assert fwd() == 0, L.r3m_last_error()
        assert dg() == 0, L.r3m_last_error()
        torch.cuda.synchronize()
        outs[mode] = (y.clone(), stats.clone(), dx.clone())
        if do_time:
            times[mode] = (timed(fwd), timed(dg))
    L.r3m_debug_set_pw16(0)
    names = ("y", "stats", "dx")
    def same(n, a, b):
        if n == "stats":      # the big-tile kernel sums row PAIRS (packed fp32 adds): fp32-level differences in the partial rows
            return bool(torch.allclose(a, b, rtol=2e-5, atol=1e-4 * float(a.abs().max())))
        return torch.equal(a.view(torch.int16), b.view(torch.int16))
    bad = [n for n, a, b in zip(names, outs[0], outs[1]) if not same(n, a, b)]
    line = f"{'x'.join(str(v) for v in case):30s} M={N * Ho * Ho:9d} " + ("IDENTICAL" if not bad else "MISMATCH " + ",".join(bad))
    if bad:
        for n, a, b in zip(names, outs[0], outs[1]):
            if n in bad:
                d = (a.float() - b.float())
                nan = int(torch.isnan(b.float()).sum())
                line += f" [{n}: max|d|={float(torch.nan_to_num(d).abs().max()):.4g} nan_new={nan} differing={int((d != 0).sum())}/{d.numel()}]"
    if do_time:
        line += f"  fwd {times[0][0]:.3f} -> {times[1][0]:.3f} ms   dgrad {times[0][1]:.3f} -> {times[1][1]:.3f} ms"
    print(line, flush=True)
    return not bad


def ops():
    ok = True
    for c in SMALL:
        ok &= run_case(c, False)
    for c in BIG:
        ok &= run_case(c, True)
    print("ALL IDENTICAL" if ok else "SOME MISMATCH")
    return 0 if ok else 1


def step():
    from r3m_amd import R3M
    from r3m_amd.trainer import Trainer
    from r3m_amd import augment
    cfgs = [("configs[2] ResNet-50 bf16 + language head, 256 clips", dict(size=50, clips=256, lang=1.0, aug=False)),
            ("configs[4] ResNet-34 bf16 rctraj, 512 clips", dict(size=34, clips=512, lang=0.0, aug=True))]
    for name, c in cfgs:
        torch.manual_seed(1)
        B = c["clips"]
        model = R3M("cuda", 1e-4, 1024, size=c["size"], l2weight=1e-5, l1weight=1e-5, langweight=c["lang"], tcnweight=1.0, l2dist=True, bs=B,
                    precision="bf16").to(DEV)
        from r3m_amd.parallel import make_network_wrapper
        net = make_network_wrapper(model, force=False)
        g = torch.Generator(device=DEV).manual_seed(1234)
        if not c["aug"]:
            frames = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=DEV, dtype=torch.int32).float()
            get = lambda: frames
        else:
            raw = torch.randint(0, 256, (B, 5, 3, 256, 256), generator=g, device=DEV, dtype=torch.int32).to(torch.uint8)
            bg = torch.Generator().manual_seed(99)
            get = lambda: augment.random_resized_crop(raw, per_clip=True, generator=bg, fused=True)
        langs = [""] * B
        if c["lang"] > 0:
            gl = torch.Generator(device=DEV).manual_seed(4321)
            langs = torch.randn((B, 768), generator=gl, device=DEV) * 0.3
        tr = Trainer(eval_freq=10 ** 9)
        for i in range(6):
            tr.update(net, (get(), langs), i)
        torch.cuda.synchronize()
        res = {0: [], 1: []}
        for rep in range(3):
            for mode in (0, 1):
                L.r3m_debug_set_pw16(MODE if mode else 0)
                for i in range(2):
                    tr.update(net, (get(), langs), i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(10):
                    m, _ = tr.update(net, (get(), langs), i)
                torch.cuda.synchronize()
                res[mode].append((time.perf_counter() - t0) * 100.0)
        L.r3m_debug_set_pw16(0)
        print(f"{name}: per-tile kernels {' '.join(f'{v:.2f}' for v in res[0])} ms   persistent {' '.join(f'{v:.2f}' for v in res[1])} ms", flush=True)
        del model, net, tr
        torch.cuda.empty_cache()
    return 0


def report():
    """per-shape conv launch times INSIDE the step (the library's own HIP-event timing), per-tile kernels vs persistent kernel"""
    import collections, csv, ctypes as C
    from r3m_amd import R3M
    from r3m_amd.trainer import Trainer
    from r3m_amd.parallel import make_network_wrapper
    size, B = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 256)
    torch.manual_seed(1)
    model = R3M("cuda", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0, l2dist=True, bs=B, precision="bf16").to(DEV)
    net = make_network_wrapper(model, force=False)
    g = torch.Generator(device=DEV).manual_seed(1234)
    frames = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=DEV, dtype=torch.int32).float()
    tr = Trainer(eval_freq=10 ** 9)
    for i in range(4):
        tr.update(net, (frames, [""] * B), i)
    torch.cuda.synchronize()
    agg = {}
    steps = 5
    for mode in (0, 1):
        L.r3m_debug_set_pw16(MODE if mode else 0)
        for i in range(2):
            tr.update(net, (frames, [""] * B), i)
        torch.cuda.synchronize()
        path = f"/tmp/pw16_launch_{mode}.csv"
        L.r3m_profile_enable(1)
        _lib.check(L.r3m_profile_dump_to(path.encode()), "dump")
        for i in range(steps):
            tr.update(net, (frames, [""] * B), i)
        torch.cuda.synchronize()
        ms, ln, fl = (C.c_double * 4)(), (C.c_longlong * 4)(), (C.c_double * 4)()
        _lib.check(L.r3m_profile_collect(ms, ln, fl), "collect")
        L.r3m_profile_enable(0)
        L.r3m_profile_dump_to(None)
        for r in csv.DictReader(open(path)):
            k = (int(r["class"]), int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]))
            a = agg.setdefault(k, {0: [0, 0.0], 1: [0, 0.0]})[mode]
            a[0] += 1
            a[1] += float(r["ms"])
    L.r3m_debug_set_pw16(0)
    print("cls         M     N     K taps n/step  ms/launch per-tile -> persistent   ms/step delta")
    tot = 0.0
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0][1]):
        n0, m0 = v[0]
        n1, m1 = v[1]
        if n0 == 0 or n1 == 0:
            continue
        d = (m1 - m0) / steps
        tot += d
        print(f"{k[0]} {k[1]:11d} {k[2]:5d} {k[3]:5d} {k[4]:3d} {n0 / steps:6.1f}   {m0 / n0:8.3f} -> {m1 / n1:8.3f}   {d:+7.3f}")
    print(f"sum of deltas per step {tot:+.3f} ms")
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        sys.exit(report())
    sys.exit(ops() if (len(sys.argv) < 2 or sys.argv[1] == "ops") else step())
